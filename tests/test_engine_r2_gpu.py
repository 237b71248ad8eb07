"""GPU: the rows the round-1 review found only shape-checked -- the CE-only engine (A22), the 750-class configuration
of BASELINE config 5 with every ablation flag, Engine.test()/_evaluate (A17 / N1) and Engine.run (config 1's plumbing)
-- against goldens captured from the IMPORTED REFERENCE (tests/golden/model_golden_r2.npz, gen_model_golden_r2.py) and
against the oracle chain on the same data."""
import io
import os
import tempfile
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

from ieee_amd._spec import state_spec
from oracle import engine as oe
from oracle import model as om
from tests.util_model import (calibrated_state, eval_loaders, generated_state, images, run2_train_loader)

pytestmark = pytest.mark.gpu

KEYS_3M = ("loss", "LossX", "LossM", "lossR", "lossN", "lossT", "accR", "accN", "accT")
KEYS_SM = ("loss_all", "loss_R", "acc_R", "loss_N", "acc_N", "loss_T", "acc_T")


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "model_golden_r2.npz"))


class FakeDM(object):
    sources = ["synthetic"]
    num_instances = 4

    def __init__(self, C, train_loader=None, test_loader=None):
        self.num_train_pids = C
        self.train_loader = train_loader or []
        self.test_loader = test_loader or {}


def build(C, loss, state, dtype=torch.float32, **flags):
    from ieee_amd.models import build_model
    m = build_model("ieee3modalPart", num_classes=C, loss=loss, pretrained=False, compute_dtype=dtype, **flags)
    m.load_state_dict(state)
    return m


def shapes(C):
    return {k: s for k, s, _ in state_spec(C)}


def batch(B, seed):
    pids = torch.arange(B) // 4
    return {"img": images(B, seed), "pid": pids, "camid": pids * 0, "impath": "", "timeid": pids * 0}


def engine_for(m, kind, dm=None, lr=1e-3, sched=None):
    from ieee_amd.engine import Image3MEngine, MultiModalImageSoftmaxEngine
    from ieee_amd.optim import build_optimizer
    opt = build_optimizer(m, optim="sgd", lr=lr, weight_decay=5e-4, momentum=0.9)
    dm = dm or FakeDM(m.num_classes)
    if kind == "margin":
        return Image3MEngine(dm, m, opt, margin=1, weight_m=1, weight_x=1, scheduler=sched(opt) if sched else None, use_gpu=True)
    return MultiModalImageSoftmaxEngine(dm, m, opt, scheduler=sched(opt) if sched else None, use_gpu=True)


def sampled_update_error(model, state, names, post_ref):
    """per tensor: max error of the 32 sampled entries of (parameter after the step - before), relative to the largest
    sampled reference update (the golden keeps sum, |sum|, L2 and 32 samples of every tensor)"""
    worst = 0.0
    mine = model.state_dict()
    for i, n in enumerate(names):
        before = state[n].double().flatten()
        idx = torch.linspace(0, before.numel() - 1, 32).long()
        upd_ref = torch.from_numpy(post_ref[i][3:]) - before[idx]
        upd_my = mine[n].double().flatten().cpu()[idx] - before[idx]
        denom = upd_ref.abs().max().item()
        if denom < 1e-12:
            assert upd_my.abs().max().item() < 1e-9, n
            continue
        worst = max(worst, (upd_my - upd_ref).abs().max().item() / denom)
    return worst


def test_softmax_engine_step_matches_reference(G):
    """MultiModalImageSoftmaxEngine.forward_backward (reference engine/image/softmax.py:81-132): summary values, logits,
    the gradient-None pattern and the parameters after the SGD step"""
    state = generated_state(shapes(171), 6)
    m = build(171, "softmax", state).train()
    out = m([x.cuda() for x in images(8, 6)])
    assert len(out) == 3
    logits = torch.stack([torch.stack(list(o)) for o in out]).reshape(18, 8, 171)
    assert np.abs(logits.detach().cpu().numpy() - G["softmax8/logits"]).max() < 1e-3
    m = build(171, "softmax", state).train()
    eng = engine_for(m, "softmax")
    s = eng.forward_backward(batch(8, 6))
    assert tuple(s) == KEYS_SM == tuple(str(k) for k in G["softmax8/summary_keys"])
    np.testing.assert_allclose([float(s[k]) for k in KEYS_SM], G["softmax8/summary"], rtol=1e-4, atol=1e-4)
    names = [str(n) for n in G["softmax8/param_names"]]
    assert [n in m._no_grad_names() for n in names] == list(G["softmax8/grad_none"])
    # noise floor of this quantity between two runs of the reference itself: ~0.14 (LABNOTES.md "Parity")
    assert sampled_update_error(m, state, names, G["softmax8/post_param_stats"]) < 0.25
    sd = m.state_dict()
    bnames = [str(n) for n in G["softmax8/buffer_names"]]
    from tests.util_model import compare_stats, stats
    compare_stats([stats(sd[n]) for n in bnames], G["softmax8/post_buffer_stats"], bnames, 1e-3, "running stats")


@pytest.mark.parametrize("kind", ["margin", "softmax"])
def test_deferred_summary_equals_the_eager_one(kind):
    """`engine.defer_summary`: forward_backward returns before the device is done and the numbers arrive on first look;
    five steps queued back to back (more than the ring of pinned read-back buffers) give, step for step, the values of
    five eager steps from the same start (fp32 parity mode; both runs are the same kernels in the same order)"""
    from ieee_amd.meters import DeferredSummary
    state = generated_state(shapes(171), 9)
    runs = []
    for defer in (False, True):
        m = build(171, "margin" if kind == "margin" else "softmax", state).train()
        eng = engine_for(m, kind)
        eng.defer_summary = defer
        got = [eng.forward_backward(batch(8, 9)) for _ in range(5)]
        assert all(isinstance(s, DeferredSummary) == defer for s in got)
        if defer:
            assert not got[-1].resolved and got[0].resolved     # the ring settled the oldest ones when it wrapped
        runs.append([[float(v) for v in s.values()] for s in got])
        assert tuple(got[0]) == (KEYS_SM if kind == "softmax" else tuple(got[0].keys()))
    np.testing.assert_array_equal(np.array(runs[0]), np.array(runs[1]))


@pytest.mark.parametrize("tag,flags", [("c750_train8", {}), ("c750_train8_noatt", dict(attention=False)),
                                       ("c750_train8_nocim", dict(interaction=False)), ("c750_train8_norem", dict(using_REM=False))])
def test_750_classes_fp32_match_reference(G, tag, flags):
    """BASELINE config 5 (Market1501-multimodal: 750 identities) through the classifiers, the 18-head CE and the 3M
    loss, full model and each ablation flag, fp32 parity mode, B = 8"""
    C = 750
    state = generated_state(shapes(C), 7)
    m = build(C, "margin", state, **flags).train()
    out = m([x.cuda() for x in images(8, 7)])
    logits = torch.stack([torch.stack(list(o)) for o in out[:3]]).reshape(18, 8, C)
    feats = torch.stack(list(out[3:]))
    assert np.abs(logits.detach().cpu().numpy() - G[tag + "/logits"]).max() < 1e-3
    assert np.abs(feats.detach().cpu().numpy() - G[tag + "/feats"]).max() < 1e-3
    m = build(C, "margin", state, **flags).train()
    s = engine_for(m, "margin").forward_backward(batch(8, 7))
    assert tuple(s) == ("loss", "LossX", "LossM", "accR", "lossR", "accN", "lossN", "accT", "lossT")
    np.testing.assert_allclose([float(s[k]) for k in KEYS_3M], G[tag + "/summary"], rtol=1e-4, atol=1e-4)
    names = [n for n, _ in m.named_parameters()]
    assert [n in m._no_grad_names() for n in names] == list(G[tag + "/grad_none"])
    if not flags:
        assert sampled_update_error(m, state, [str(n) for n in G[tag + "/param_names"]], G[tag + "/post_param_stats"]) < 0.25


def test_750_classes_softmax_leg_matches_reference(G):
    state = generated_state(shapes(750), 7)
    m = build(750, "softmax", state).train()
    s = engine_for(m, "softmax").forward_backward(batch(8, 7))
    np.testing.assert_allclose([float(s[k]) for k in KEYS_SM], G["c750_softmax8/summary"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("leg", ["full", "noatt", "nocim", "norem", "3m_off"])
def test_config5_bf16_b32_ablation_sweep(leg):
    """config 5 as it is run: 32 triples per rank, 750 classes, bf16, one leg per ablation.  bf16 on this random-init net
    cannot be held to fp32 values (LABNOTES.md "Parity"), so the bar is the one of the headline configuration: the step's
    loss within 2 % of the fp32 parity mode on the same inputs, finite gradients everywhere, parameters moved."""
    C, B = 750, 32
    flags = {"noatt": dict(attention=False), "nocim": dict(interaction=False), "norem": dict(using_REM=False)}.get(leg, {})
    kind = "softmax" if leg == "3m_off" else "margin"
    state = generated_state(shapes(C), 7)
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        m = build(C, kind, state, dtype=dt, **flags).train()
        before = m._flat_params.clone()
        s = engine_for(m, kind).forward_backward(batch(B, 7))
        torch.cuda.synchronize()
        res[dt] = float(s["loss_all" if kind == "softmax" else "loss"])
        assert torch.isfinite(m._flat_grads).all() and torch.isfinite(m._flat_params).all()
        assert not torch.equal(before, m._flat_params)
        del m
    assert abs(res[torch.bfloat16] - res[torch.float32]) / res[torch.float32] < 0.02, res


def test_engine_test_matches_reference_evaluate(G):
    """Engine.test() -> _evaluate on tiny query / gallery loaders (reference engine/engine.py:287-441): descriptors,
    CMC curve, mAP and the printed report against the reference's own CPU run AND the oracle chain on the same data"""
    import ieee_amd.engine as E
    state = calibrated_state(shapes(171), 8)
    m = build(171, "margin", state)
    L = eval_loaders()
    eng = engine_for(m, "margin", FakeDM(171, test_loader={"synthetic": L}))
    seen = {}
    orig = E.evaluate_rank

    def spy(distmat, *a, **k):
        cmc, m_ap = orig(distmat, *a, **k)
        seen["distmat"], seen["cmc"], seen["mAP"] = distmat, np.asarray(cmc), m_ap
        return cmc, m_ap
    E.evaluate_rank = spy
    try:
        with redirect_stdout(io.StringIO()) as buf:
            m_ap = eng.test()
    finally:
        E.evaluate_rank = orig
    assert not m.training
    # against the reference's run
    assert np.array_equal(seen["cmc"], G["evalpipe/cmc"])
    assert abs(m_ap - float(G["evalpipe/mAP"])) < 1e-9 and m_ap == seen["mAP"]
    dm = seen["distmat"].cpu().numpy() if torch.is_tensor(seen["distmat"]) else np.asarray(seen["distmat"])
    np.testing.assert_allclose(dm, G["evalpipe/distmat"], rtol=2e-4, atol=5e-2)
    ref_lines = [l for l in str(G["evalpipe/printed"]).splitlines() if l.startswith(("mAP", "Rank-", "CMC", "** Results"))]
    my_lines = [l for l in buf.getvalue().splitlines() if l.startswith(("mAP", "Rank-", "CMC", "** Results"))]
    assert my_lines == ref_lines and any(l.startswith("Rank-20") for l in my_lines)
    # against the oracle chain (eval forward on the CPU -> sgemm-form distmat -> the C evaluator) on the same loaders
    cmc_o, map_o, qf_o, gf_o, dist_o = oe.evaluate(state, L["query"], L["gallery"])
    assert np.array_equal(seen["cmc"], cmc_o) and abs(m_ap - map_o) < 1e-9
    np.testing.assert_allclose(dm, dist_o, rtol=2e-4, atol=5e-2)


def _engine_test_spied(state, L, dtype):
    """Engine.test() of a native model in `dtype` on loaders L -> (returned mAP, cmc, qf, gf)"""
    import ieee_amd.engine as E
    m = build(171, "margin", state, dtype=dtype)
    eng = engine_for(m, "margin", FakeDM(171, test_loader={"synthetic": L}))
    seen = {}
    orig_rank, orig_dist = E.evaluate_rank, E.compute_distance_matrix

    def spy_rank(distmat, *a, **k):
        cmc, m_ap = orig_rank(distmat, *a, **k)
        seen["cmc"] = np.asarray(cmc)
        return cmc, m_ap

    def spy_dist(qf, gf, *a, **k):
        seen["qf"], seen["gf"] = qf.detach().float().cpu(), gf.detach().float().cpu()
        return orig_dist(qf, gf, *a, **k)
    E.evaluate_rank, E.compute_distance_matrix = spy_rank, spy_dist
    try:
        with redirect_stdout(io.StringIO()):
            m_ap = eng.test()
    finally:
        E.evaluate_rank, E.compute_distance_matrix = orig_rank, orig_dist
    return m_ap, seen["cmc"], seen["qf"], seen["gf"]


def _stock_bf16(state, L):
    """stock torch bf16 autocast of the oracle's eval forward on the device -> (mAP, cmc, qf, gf): the yardstick"""
    from oracle import evaluator as ev
    sd = {k: v.cuda() for k, v in state.items()}

    def feats(loader):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return torch.cat([om.forward({k: v.clone() for k, v in sd.items()}, [x.cuda() for x in d["img"]], False).float().cpu()
                              for d in loader], 0)
    sq, sg = feats(L["query"]), feats(L["gallery"])
    lab = lambda loader, key: np.concatenate([np.asarray(d[key]) for d in loader])
    cmc, m_ap = ev.rank_market1501_c(ev.sqeuclid_np(sq.numpy(), sg.numpy()), lab(L["query"], "pid"), lab(L["gallery"], "pid"),
                                     lab(L["query"], "camid"), lab(L["gallery"], "camid"))
    return m_ap, cmc, sq, sg


@pytest.mark.parametrize("fixture", ["evalpipe", "evalpipe_tame", "evalpipe_tame_hard"])
def test_engine_test_in_bf16_matches_reference_evaluate(G, golden_dir, fixture):
    """The drop-in DEFAULT evaluation path -- compute_dtype = bf16: 53 BN-folded bf16 convs per stream, the CIM, the fp32
    head, descriptors -> distmat -> ranking -- against the REFERENCE's own fp32 CPU run of Engine.test() on the same loaders.

    `evalpipe_tame` / `evalpipe_tame_hard` (tests/golden/model_golden_r3.npz; every bottleneck's last BatchNorm scale
    x 0.25, so the net is not chaotic: bf16 drift ~10 %): first the fp32 parity mode reproduces the reference (CMC
    bit-equal, mAP 1e-9) -- the fixture is sound -- then bf16: descriptors no further from the reference's than stock
    torch bf16 autocast x 1.25, and on `tame` (identities separate: reference mAP = 1) mAP within 1e-3 and the CMC curve
    within one query; on `tame_hard` (mAP 0.918, near-ties at the bf16 noise level) and on round 2's chaotic `evalpipe`
    (bf16 drift 60-80 % for ANY implementation) no further from the reference than stock bf16 is (x 1.25, + one query)."""
    tame = fixture != "evalpipe"
    R = np.load(os.path.join(golden_dir, "model_golden_r3.npz")) if tame else G
    noise = float(R[fixture + "/noise"]) if tame else 0.5
    state = calibrated_state(shapes(171), 8, tame=tame, noise=noise)
    L = eval_loaders(noise)
    ref_map, ref_cmc = float(R[fixture + "/mAP"]), R[fixture + "/cmc"]
    ref_q, ref_g = torch.from_numpy(R[fixture + "/qf"]), torch.from_numpy(R[fixture + "/gf"])
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    if tame:        # the fp32 parity mode on the new fixture: the reference's result, exactly
        m32, c32, q32, g32 = _engine_test_spied(state, L, torch.float32)
        assert np.array_equal(c32, ref_cmc) and abs(m32 - ref_map) < 1e-9
        assert rel(g32, ref_g) < 1e-4 and rel(q32, ref_q) < 1e-4
    m16, c16, q16, g16 = _engine_test_spied(state, L, torch.bfloat16)
    ms, cs, sq, sg = _stock_bf16(state, L)
    e_g, e_q, s_g, s_q = rel(g16, ref_g), rel(q16, ref_q), rel(sg, ref_g), rel(sq, ref_q)
    dev, sdev = float(np.abs(c16 - ref_cmc).max()), float(np.abs(cs - ref_cmc).max())
    print("%s bf16 eval: mAP native %.6f / stock torch bf16 %.6f / reference %.6f; max CMC deviation native %.4f / stock %.4f; "
          "descriptor error vs the reference: gallery %.3e / stock %.3e, query %.3e / stock %.3e"
          % (fixture, m16, ms, ref_map, dev, sdev, e_g, s_g, e_q, s_q))
    assert e_g <= 1.25 * s_g + 1e-3 and e_q <= 1.25 * s_q + 1e-3
    nq = ref_q.shape[0]
    if fixture == "evalpipe_tame":
        assert abs(m16 - ref_map) <= 1e-3 and dev <= 1.0 / nq + 1e-6
    else:
        assert abs(m16 - ref_map) <= 1.25 * abs(ms - ref_map) + 1.0 / nq, (m16, ms, ref_map)
        assert dev <= sdev + 2.0 / nq + 1e-6, (dev, sdev)


def test_engine_run_matches_reference_loop(G):
    """Engine.run(max_epoch=2, eval_freq=1) on a synthetic datamanager: per-batch summaries, the learning-rate schedule,
    the evaluation + checkpoint after epoch 1 only (none after the last epoch, engine.py:216), counters"""
    from ieee_amd.checkpoint import resume_from_checkpoint
    from ieee_amd.optim import build_lr_scheduler
    state = calibrated_state(shapes(171), 9)
    m = build(171, "margin", state)
    dm = FakeDM(171, train_loader=run2_train_loader(), test_loader={"synthetic": eval_loaders()})
    eng = engine_for(m, "margin", dm, sched=lambda opt: build_lr_scheduler(opt, "multi_step", stepsize=[1], gamma=0.1))
    summaries, evals = [], []
    fb, ev = eng.forward_backward, eng._evaluate

    def fb_rec(data):
        s = fb(data)
        summaries.append([float(s[k]) for k in KEYS_3M])
        return s

    def ev_rec(**k):
        r1, m_ap = ev(**k)
        evals.append([float(r1), float(m_ap)])
        return r1, m_ap
    eng.forward_backward, eng._evaluate = fb_rec, ev_rec
    with tempfile.TemporaryDirectory() as d, redirect_stdout(io.StringIO()) as buf:
        eng.run(save_dir=d, max_epoch=2, eval_freq=1, print_freq=1)
        saved = sorted(os.listdir(os.path.join(d, "model")))
        assert saved == [str(x) for x in G["run2/saved"]] == ["model.pth.tar-1"]
        m2 = build(171, "margin", generated_state(shapes(171), 1))
        assert resume_from_checkpoint(os.path.join(d, "model", saved[0]), m2) == 1
    got, ref = np.array(summaries), G["run2/summaries"]
    assert got.shape == ref.shape == (4, 9)
    # first step: fp32 parity-mode accuracy.  Every later step sees weights that carry the gradient noise floor of the
    # updates before it (a flipped ReLU mask changes a gradient by O(|g|); with 4 rows per batch the oracle against ITSELF
    # with another thread count already differs by 1e-3 on the step-3 loss, tests/test_engine_oracle.py; measured here:
    # 1.4e-3 on step 2)
    np.testing.assert_allclose(got[0, :6], ref[0, :6], rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(got[1:, :6], ref[1:, :6], rtol=1e-2, atol=1e-3)
    np.testing.assert_allclose(got[:, 6:], ref[:, 6:], atol=100.0 / 24 + 1e-6)
    np.testing.assert_allclose(evals, G["run2/evals"], atol=1e-9)
    assert abs(eng.get_current_lr() - float(G["run2/final_lr"])) < 1e-12
    out = buf.getvalue()
    assert "=> Start training" in out and "epoch: [2/2][2/2]" in out and out.count("##### Evaluating synthetic (source) #####") == 1
    sd = m.state_dict()
    assert np.array_equal(np.array([int(sd[k]) for k in sd if k.endswith("num_batches_tracked")]), G["run2/nbt"])


def test_run_test_only_and_rerank_flag():
    """run(test_only=True, rerank=True) evaluates through the k-reciprocal re-ranking (N3) instead of refusing it; the
    3M engine raises the reference's IndexError BEFORE the weights change when chunk() yields fewer pieces than identities"""
    state = calibrated_state(shapes(171), 8)
    m = build(171, "margin", state)
    eng = engine_for(m, "margin", FakeDM(171, test_loader={"synthetic": eval_loaders()}))
    with redirect_stdout(io.StringIO()) as buf:
        eng.run(test_only=True, rerank=True)
    assert "Applying person re-ranking" in buf.getvalue() and "mAP:" in buf.getvalue()
    m.train()
    before = m._flat_params.clone()
    bad = batch(8, 3)
    bad["pid"] = torch.tensor([0, 0, 0, 1, 1, 2, 2, 2])       # 3 identities in 8 rows: chunk(3) -> pieces of 3, 3, 2 = 3 OK
    eng.forward_backward(bad)
    assert not torch.equal(before, m._flat_params)
    before = m._flat_params.clone()
    bad["pid"] = torch.tensor([0, 1, 2, 3, 4, 0, 1, 2])       # 5 identities: ceil(8/5) = 2 rows per piece -> only 4 pieces
    with pytest.raises(IndexError):
        eng.forward_backward(bad)
    assert torch.equal(before, m._flat_params)
