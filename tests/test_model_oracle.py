"""CPU: the model oracle (oracle/model.py, stock torch fp32 ops) against the golden vectors captured
from the imported reference (tests/golden/gen_model_golden.py)."""
import os

import numpy as np
import pytest
import torch

from ieee_amd._spec import state_spec
from oracle import model as om
from tests.util_model import C, compare_stats, generated_state, images, stats


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "model_golden.npz"))


def shapes():
    return {k: s for k, s, _ in state_spec(C)}


def test_spec_matches_reference_state_dict(golden_dir):
    ref = [l.split() for l in open(os.path.join(golden_dir, "state_dict_spec.txt")) if not l.startswith("#")]
    mine = [(k, ",".join(map(str, s))) for k, s, _ in state_spec(C)]
    assert [(r[0], r[1] if len(r) > 1 else "") for r in ref] == mine


def test_oracle_eval_forward_matches_reference(G):
    torch.set_num_threads(8)
    sd = generated_state(shapes(), seed=1)
    xs = images(4, seed=1)
    with torch.no_grad():
        fc = om.forward(sd, xs, False)
        assert np.abs(fc.numpy() - G["eval/fc_all"]).max() < 1e-4
        fc = om.forward(sd, xs, False, attention=False)
        assert np.abs(fc.numpy() - G["eval_noatt/fc_all"]).max() < 1e-4


def test_oracle_train_step_matches_reference(G):
    torch.set_num_threads(8)
    sd = generated_state(shapes(), seed=3)
    xs = images(8, seed=3)
    pids = torch.arange(8) // 4
    summary, grads, new_sd, _ = om.train_step(sd, xs, pids, C)
    ref = G["train8/summary"]
    keys = ("loss", "LossX", "LossM", "lossR", "lossN", "lossT", "accR", "accN", "accT")
    np.testing.assert_allclose([summary[k] for k in keys], ref, rtol=1e-4, atol=1e-4)
    names = [str(n) for n in G["train8/param_names"]]
    none_ref = G["train8/grad_none"]
    assert [grads[n] is None for n in names] == list(none_ref)
    mine = [stats(grads[n]) if grads[n] is not None else np.zeros(35) for n in names]
    compare_stats(mine, G["train8/grad_stats"], names, 2e-3, "oracle grads")
    compare_stats([stats(new_sd[n]) for n in names], G["train8/post_param_stats"], names, 1e-5, "post-SGD params")
    bnames = [str(n) for n in G["train8/buffer_names"]]
    compare_stats([stats(new_sd[n]) for n in bnames], G["train8/post_buffer_stats"], bnames, 1e-4, "running stats")


@pytest.mark.parametrize("tag,seed", [("fix_cls", 6), ("fix_backbone", 7)])
def test_oracle_frozen_children_match_the_reference_two_stepped_transfer_learning(golden_dir, tag, seed):
    """tests/golden/gen_model_golden_r4.py: Engine.two_stepped_transfer_learning(0, 1, open_layers) on the imported reference
    (engine.py:507-529, utils/torchtools.py:183-221), one step under the freeze, then a step with everything open"""
    torch.set_num_threads(8)
    G4 = np.load(os.path.join(golden_dir, "model_golden_r4.npz"))
    keep = [str(n) for n in G4[tag + "/open_layers"]]
    children = []
    for k, _, _ in state_spec(C):
        if k.split(".")[0] not in children:
            children.append(k.split(".")[0])
    frozen = tuple(c for c in children if c not in keep)
    sd = generated_state(shapes(), seed=seed)
    xs = images(8, seed=seed)
    pids = torch.arange(8) // 4
    summary, grads, new_sd, mom = om.train_step(sd, xs, pids, C, frozen=frozen)
    keys = ("loss", "LossX", "LossM", "lossR", "lossN", "lossT", "accR", "accN", "accT")
    np.testing.assert_allclose([summary[k] for k in keys], G4[tag + "/summary"], rtol=2e-4, atol=1e-4)
    names = [str(n) for n in G4[tag + "/param_names"]]
    assert [grads.get(n) is None for n in names] == list(G4[tag + "/grad_none"])
    mine = [stats(grads[n]) if grads.get(n) is not None else np.zeros(35) for n in names]
    compare_stats(mine, G4[tag + "/grad_stats"], names, 2e-3, "oracle grads under the freeze")
    compare_stats([stats(new_sd[n]) for n in names], G4[tag + "/post_param_stats"], names, 1e-5, "post-SGD params")
    assert [bool(torch.equal(new_sd[n], sd[n])) for n in names] == list(G4[tag + "/param_unchanged"])
    bnames = [str(n) for n in G4[tag + "/buffer_names"]]
    compare_stats([stats(new_sd[n]) for n in bnames], G4[tag + "/post_buffer_stats"], bnames, 1e-4, "running stats")
    assert [bool(torch.equal(new_sd[n], sd[n])) for n in bnames] == list(G4[tag + "/buffer_unchanged"])
    # the freeze ends: the second step trains everything (momentum state of the first step carried along)
    summary2, grads2, _, _ = om.train_step(new_sd, xs, pids, C, mom_state=mom)
    np.testing.assert_allclose([summary2[k] for k in keys], G4[tag + "/summary_step2"], rtol=1e-3, atol=1e-3)
    assert [grads2[n] is None for n in names] == list(G4[tag + "/step2_grad_none"])
