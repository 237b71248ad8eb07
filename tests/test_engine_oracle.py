"""CPU: the oracle's restatements added in round 2 -- the CE-only engine step, the 750-class configuration, the
evaluation chain (eval forward -> squared-Euclidean distmat -> Market-1501 CMC / mAP) and the Engine.run loop --
against goldens captured from the IMPORTED REFERENCE (tests/golden/gen_model_golden_r2.py)."""
import os

import numpy as np
import pytest
import torch

from ieee_amd._spec import state_spec
from oracle import engine as oe
from oracle import model as om
from tests.util_model import (calibrated_state, compare_stats, eval_loaders, generated_state, images, run2_train_loader,
                              stats)

KEYS_3M = ("loss", "LossX", "LossM", "lossR", "lossN", "lossT", "accR", "accN", "accT")
KEYS_SM = ("loss_all", "loss_R", "acc_R", "loss_N", "acc_N", "loss_T", "acc_T")


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "model_golden_r2.npz"))


def shapes(C):
    return {k: s for k, s, _ in state_spec(C)}


def test_oracle_softmax_engine_step_matches_reference(G):
    torch.set_num_threads(8)
    sd = generated_state(shapes(171), seed=6)
    pids = torch.arange(8) // 4
    summary, grads, new_sd, _ = om.train_step(sd, images(8, 6), pids, 171, engine="softmax")
    assert tuple(str(k) for k in G["softmax8/summary_keys"]) == KEYS_SM
    np.testing.assert_allclose([summary[k] for k in KEYS_SM], G["softmax8/summary"], rtol=1e-4, atol=1e-4)
    names = [str(n) for n in G["softmax8/param_names"]]
    assert [grads[n] is None for n in names] == list(G["softmax8/grad_none"])
    compare_stats([stats(grads[n]) if grads[n] is not None else np.zeros(35) for n in names], G["softmax8/grad_stats"],
                  names, 2e-3, "oracle grads (CE-only engine)")
    compare_stats([stats(new_sd[n]) for n in names], G["softmax8/post_param_stats"], names, 1e-5, "post-SGD params")


@pytest.mark.parametrize("tag,flags", [("c750_train8", {}), ("c750_train8_nocim", dict(interaction=False))])
def test_oracle_750_classes_match_reference(G, tag, flags):
    torch.set_num_threads(8)
    sd = generated_state(shapes(750), seed=7)
    pids = torch.arange(8) // 4
    summary, grads, _, _ = om.train_step(sd, images(8, 7), pids, 750, **flags)
    np.testing.assert_allclose([summary[k] for k in KEYS_3M], G[tag + "/summary"], rtol=1e-4, atol=1e-4)
    names = [n for n in grads]
    assert [grads[n] is None for n in names] == list(G[tag + "/grad_none"])


def test_oracle_evaluation_chain_matches_reference(G):
    torch.set_num_threads(8)
    sd = calibrated_state(shapes(171), seed=8)
    bn = torch.cat([v.flatten() for k, v in sd.items() if "running_" in k])
    np.testing.assert_allclose(stats(bn)[:3], G["evalpipe/bn_check"][:3], rtol=1e-5)
    L = eval_loaders()
    cmc, m_ap, qf, gf, dist = oe.evaluate(sd, L["query"], L["gallery"])
    scale = np.abs(G["evalpipe/gf"]).max()
    assert np.abs(qf - G["evalpipe/qf"]).max() < 1e-4 * scale and np.abs(gf - G["evalpipe/gf"]).max() < 1e-4 * scale
    np.testing.assert_allclose(dist, G["evalpipe/distmat"], rtol=1e-4, atol=1e-2)
    assert np.array_equal(cmc, G["evalpipe/cmc"])
    assert abs(m_ap - float(G["evalpipe/mAP"])) < 1e-9
    assert float(G["evalpipe/returned_mAP"]) == float(G["evalpipe/mAP"])


@pytest.mark.parametrize("fixture", ["evalpipe_tame", "evalpipe_tame_hard"])
def test_oracle_evaluation_chain_matches_reference_on_the_tame_fixtures(golden_dir, fixture):
    """round 3: the well-conditioned evaluation fixtures of the bf16 tests (tests/golden/gen_model_golden_r3.py: every
    bottleneck's last BatchNorm scale x 0.25) -- the oracle chain reproduces the reference's descriptors, CMC and mAP"""
    torch.set_num_threads(8)
    R = np.load(os.path.join(golden_dir, "model_golden_r3.npz"))
    noise = float(R[fixture + "/noise"])
    sd = calibrated_state(shapes(171), seed=8, tame=True, noise=noise)
    L = eval_loaders(noise)
    cmc, m_ap, qf, gf, dist = oe.evaluate(sd, L["query"], L["gallery"])
    scale = np.abs(R[fixture + "/gf"]).max()
    assert np.abs(qf - R[fixture + "/qf"]).max() < 1e-4 * scale and np.abs(gf - R[fixture + "/gf"]).max() < 1e-4 * scale
    assert np.array_equal(cmc, R[fixture + "/cmc"]) and abs(m_ap - float(R[fixture + "/mAP"])) < 1e-9


def test_oracle_run_loop_matches_reference(G):
    """2 epochs x 2 batches, MultiStepLR([1]), evaluation + checkpoint after epoch 1 only (never after the last epoch)"""
    torch.set_num_threads(8)
    sd = calibrated_state(shapes(171), seed=9)
    sd, summaries, evals, saved, lr = oe.run(sd, run2_train_loader(), 171, max_epoch=2, eval_freq=1, lr=1e-3, milestones=(1,),
                                            test_loaders=eval_loaders())
    ref = G["run2/summaries"]
    got = np.array([[s[k] for k in KEYS_3M] for s in summaries])
    # the first epoch agrees to rounding; later steps inherit the gradient noise floor of the earlier ones through the
    # updated weights (a flipped ReLU mask changes a gradient by O(|g|): LABNOTES.md "Parity"), measured 6e-4 here
    np.testing.assert_allclose(got[:2, :6], ref[:2, :6], rtol=2e-5, atol=1e-4)
    np.testing.assert_allclose(got[2:, :6], ref[2:, :6], rtol=4e-3, atol=1e-3)
    np.testing.assert_allclose(got[:2, 6:], ref[:2, 6:], atol=1e-6)
    np.testing.assert_allclose(got[2:, 6:], ref[2:, 6:], atol=100.0 / 24 + 1e-6)      # at most one of the 24 head-rows flips
    assert [e[0] for e in evals] == [0] and saved == [1] and [str(x) for x in G["run2/saved"]] == ["model.pth.tar-1"]
    np.testing.assert_allclose([evals[0][1], evals[0][2]], G["run2/evals"][0], atol=1e-6)
    assert abs(lr - float(G["run2/final_lr"])) < 1e-12
    names = [str(n) for n in G["run2/param_names"]]
    # four chaotic steps apart (the oracle against ITSELF with 3 instead of 8 threads: 4.7 % of the tensor's scale on the
    # stem BatchNorm biases, 1e-3 on the step-3 loss), so this only catches a wrong schedule / momentum / decay
    compare_stats([stats(sd[n]) for n in names], G["run2/final_param_stats"], names, 3e-2, "parameters after the run")
    # (num_batches_tracked is bookkeeping the functional oracle does not carry; the GPU test checks it against the golden)
