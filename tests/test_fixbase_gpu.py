"""GPU: two-stepped transfer learning (reference Engine.two_stepped_transfer_learning, torchreid/engine/engine.py:507-529 ->
open_specified_layers / open_all_layers, utils/torchtools.py:160-221) through the native engine in the fp32 parity mode,
against goldens captured from the imported reference (tests/golden/gen_model_golden_r4.py): during the freeze every child
outside `open_layers` normalises with its running statistics, leaves them and its parameters alone, and the gradient
still reaches the open children through the frozen BatchNorms."""
import io
import os
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

from tests.util_model import C, compare_stats, generated_state, images, stats

pytestmark = pytest.mark.gpu
KEYS = ("loss", "LossX", "LossM", "lossR", "lossN", "lossT", "accR", "accN", "accT")


class _DM(object):
    num_train_pids = C
    train_loader, test_loader, sources = [], {}, ["synthetic"]


@pytest.mark.parametrize("tag,seed", [("fix_cls", 6), ("fix_backbone", 7)])
def test_frozen_children_match_the_reference_step(golden_dir, tag, seed):
    from ieee_amd.engine import Image3MEngine
    from ieee_amd.models import build_model
    from ieee_amd.optim import build_optimizer
    G = np.load(os.path.join(golden_dir, "model_golden_r4.npz"))
    keep = [str(n) for n in G[tag + "/open_layers"]]
    m = build_model("ieee3modalPart", num_classes=C, loss="margin", pretrained=False, compute_dtype=torch.float32)
    m.load_state_dict(generated_state({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed))
    eng = Image3MEngine(_DM(), m, build_optimizer(m, optim="sgd", lr=1e-3, weight_decay=5e-4, momentum=0.9), margin=1,
                        weight_m=1, weight_x=1, use_gpu=True, label_smooth=True)
    xs = images(8, seed)
    pids = torch.arange(8) // 4
    batch = lambda: {"img": [x.clone() for x in xs], "pid": pids, "camid": pids * 0, "impath": "", "timeid": pids * 0}
    eng.set_model_mode("train")
    with redirect_stdout(io.StringIO()) as said:
        eng.two_stepped_transfer_learning(0, 1, keep)
    assert "* Only train" in said.getvalue()
    names = [str(n) for n in G[tag + "/param_names"]]
    params = dict(m.named_parameters())
    frozen_ref = G[tag + "/grad_none"]
    # what the freeze covers: exactly the parameters the reference leaves without a gradient, minus the ones that never get
    # one (REM.conv_value)
    assert [not params[n].requires_grad or ".conv_value." in n for n in names] == list(frozen_ref)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    s1 = eng.forward_backward(batch())
    ref = G[tag + "/summary"]
    # (fix_cls evaluates random-init trunks under GENERATED running statistics: activations of 1e3 and a loss of 3e3)
    np.testing.assert_allclose([float(s1[k]) for k in KEYS], ref, rtol=1e-3, atol=1e-3)
    got = [stats(m._flat_grads[m._offsets[n]:m._offsets[n] + params[n].numel()]) if not frozen_ref[i] else np.zeros(35)
           for i, n in enumerate(names)]
    compare_stats(got, G[tag + "/grad_stats"], names, 3e-2, "gradients under the freeze")
    sd = m.state_dict()
    assert [bool(torch.equal(sd[n], before[n])) for n in names] == list(G[tag + "/param_unchanged"])
    # the updates of the open children: every sampled entry of (parameter after - before) against the reference's, relative to
    # the tensor's largest sampled update -- the bar of the other engine tests (the ReLU-flip noise floor of a gradient on
    # these inputs, LABNOTES.md section 4, is a few per cent of the tensor's scale)
    from tests.test_engine_r2_gpu import sampled_update_error
    state = {k: v.cpu() for k, v in before.items()}
    assert sampled_update_error(m, state, names, G[tag + "/post_param_stats"]) < 0.25
    bnames = [str(n) for n in G[tag + "/buffer_names"]]
    assert [bool(torch.equal(sd[n], before[n])) for n in bnames] == list(G[tag + "/buffer_unchanged"])
    compare_stats([stats(sd[n]) for n in bnames], G[tag + "/post_buffer_stats"], bnames, 1e-3, "running statistics")
    assert np.array_equal(np.array([int(sd[k]) for k in sd if k.endswith("num_batches_tracked")]), G[tag + "/nbt"])
    # the freeze ends: everything trains
    eng.set_model_mode("train")
    eng.two_stepped_transfer_learning(1, 1, keep)
    assert all(p.requires_grad for p in m.parameters()) and m._frozen_mask == 0
    s2 = eng.forward_backward(batch())
    # second step: the parameters already carry the first step's gradient noise (ReLU flips on this untamed random-init net:
    # LABNOTES.md section 4 -- the reference against ITSELF with another thread count moves a step-3 loss by 1e-3 and single
    # tensors by 5 %); measured here: total loss 0.6 %, one modality's cross entropy 1.3 % from the reference.  Losses to
    # 3 %, accuracies to three (sample, head) flips of the 8 x 6 that make one modality's number.
    got2, want2 = np.array([float(s2[k]) for k in KEYS]), G[tag + "/summary_step2"]
    np.testing.assert_allclose(got2[:6], want2[:6], rtol=3e-2, atol=1e-2)
    assert np.abs(got2[6:] - want2[6:]).max() <= 3 * 100.0 / 48 + 1e-6
    sd = m.state_dict()
    assert np.array_equal(np.array([int(sd[k]) for k in sd if k.endswith("num_batches_tracked")]), G[tag + "/nbt_step2"])


def test_frozen_backward_in_bf16_tracks_the_fp32_parity_mode():
    """the bf16 forms of the frozen BatchNorm backward (ieee_bn2d_bwd_frozen behind the fused dgrad epilogues): same freeze,
    tamed net, bf16 against fp32 on the gradients that reach the open backbone"""
    from ieee_amd.engine import Image3MEngine
    from ieee_amd.models import build_model
    from ieee_amd.optim import build_optimizer
    from tests.util_model import tame_
    grads = {}
    xs = images(8, 7)
    pids = torch.arange(8) // 4
    for dt in (torch.float32, torch.bfloat16):
        m = build_model("ieee3modalPart", num_classes=C, loss="margin", pretrained=False, compute_dtype=dt)
        st = tame_(generated_state({k: tuple(v.shape) for k, v in m.state_dict().items()}, 7))
        m.load_state_dict(st)
        eng = Image3MEngine(_DM(), m, build_optimizer(m, optim="sgd", lr=0.0, weight_decay=0.0, momentum=0.0), margin=1, use_gpu=True)
        eng.set_model_mode("train")
        with redirect_stdout(io.StringIO()):
            eng.two_stepped_transfer_learning(0, 1, ["backbone"])
        s = eng.forward_backward({"img": xs, "pid": pids, "camid": pids * 0, "impath": "", "timeid": pids * 0})
        runs = m.trainable_runs()
        grads[dt] = (float(s["loss"]), torch.cat([m._flat_grads[a:b] for a, b in runs]).double().cpu())
        del eng, m
        torch.cuda.empty_cache()
    (l32, g32), (l16, g16) = grads[torch.float32], grads[torch.bfloat16]
    cos = float((g32 * g16).sum() / (g32.norm() * g16.norm()))
    print("frozen head, open backbone: loss fp32 %.4f bf16 %.4f, cosine of the backbone gradients %.4f, norm ratio %.4f" % (
        l32, l16, cos, float(g16.norm() / g32.norm())))
    assert abs(l16 - l32) < 2e-2 * abs(l32)
    assert cos > 0.7 and 0.7 < float(g16.norm() / g32.norm()) < 1.4
