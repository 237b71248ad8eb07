"""Round-4 golden vectors: the reference's two-stepped transfer learning (Engine.two_stepped_transfer_learning,
torchreid/engine/engine.py:507-529 -> open_specified_layers / open_all_layers, utils/torchtools.py:160-221): during the first
`fixbase_epoch` epochs every child outside `open_layers` is in eval() mode (its BatchNorms normalise with the running
statistics and do not update them) and receives no gradient; afterwards everything trains.

  fix_cls       open_layers = the three classifier lists: everything below runs frozen
  fix_backbone  open_layers = ['backbone']: the trunks train UNDER a frozen CIM / reduce layer / REM / fc head, so the
                gradient reaches them through frozen BatchNorms (2-d and 1-d)
Each: one Image3MEngine.forward_backward on the CPU in fp32 under the freeze (summary, logits, features, which gradients are
None, gradient / post-SGD parameter / running-statistic checksums, num_batches_tracked), then open_all_layers and a second
step on the same batch (summary).
Run (this container only, needs /root/reference):  python tests/golden/gen_model_golden_r4.py"""
import io
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.ref_import import import_reference  # noqa: E402
from ieee_amd import detgen  # noqa: E402

import_reference()
from torchreid.models import build_model  # noqa: E402
from torchreid.engine import Image3MEngine  # noqa: E402

torch.set_num_threads(8)
C, B, K = 171, 8, 4
out = {}
KEYS = ("loss", "LossX", "LossM", "lossR", "lossN", "lossT", "accR", "accN", "accT")


class FakeDM:
    num_train_pids = C
    train_loader = []
    test_loader = {}
    sources = ["synthetic"]


def stats(t):
    t = t.detach().double().flatten()
    idx = torch.linspace(0, t.numel() - 1, 32).long()
    return np.concatenate([[t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()], t[idx].numpy()])


def case(tag, open_layers, seed):
    m = build_model("ieee3modalPart", num_classes=C, loss="margin", pretrained=False, use_gpu=False)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in detgen.generate_state(shapes, seed=seed).items()})
    xs = [torch.from_numpy(x) for x in detgen.generate_images(B, seed=seed)]
    pids = torch.arange(B) // K
    opt = torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9, weight_decay=5e-4, dampening=0, nesterov=True)
    eng = Image3MEngine(FakeDM(), m, opt, margin=1, weight_m=1, weight_x=1, use_gpu=False, label_smooth=True)
    batch = lambda: {"img": [x.clone() for x in xs], "pid": pids, "camid": pids * 0, "impath": "", "timeid": pids * 0}
    eng.set_model_mode("train")
    with redirect_stdout(io.StringIO()):
        eng.two_stepped_transfer_learning(0, 1, open_layers)
    captured = {}
    orig = m.forward

    def fwd(*a, **k):
        o = orig(*a, **k)
        captured["out"] = o
        return o
    m.forward = fwd
    before = {k: v.clone() for k, v in m.state_dict().items()}
    s1 = eng.forward_backward(batch())
    m.forward = orig
    oR, oN, oT, fR, fN, fT = captured["out"]
    res = {"logits": torch.stack([torch.stack(list(o)) for o in (oR, oN, oT)]).detach().numpy().reshape(18, B, C),
           "feats": torch.stack([fR, fN, fT]).detach().numpy(),
           "summary": np.array([float(s1[k]) for k in KEYS]), "open_layers": np.array(open_layers)}
    names = [k for k, _ in m.named_parameters()]
    res["param_names"] = np.array(names)
    res["grad_none"] = np.array([p.grad is None for _, p in m.named_parameters()])
    res["grad_stats"] = np.stack([stats(p.grad) if p.grad is not None else np.zeros(35) for _, p in m.named_parameters()])
    sd = m.state_dict()
    res["post_param_stats"] = np.stack([stats(sd[k]) for k in names])
    res["param_unchanged"] = np.array([bool(torch.equal(sd[k], before[k])) for k in names])
    bn_keys = [k for k in sd if k.endswith("running_mean") or k.endswith("running_var")]
    res["buffer_names"] = np.array(bn_keys)
    res["post_buffer_stats"] = np.stack([stats(sd[k]) for k in bn_keys])
    res["buffer_unchanged"] = np.array([bool(torch.equal(sd[k], before[k])) for k in bn_keys])
    res["nbt"] = np.array([int(sd[k]) for k in sd if k.endswith("num_batches_tracked")])
    # the freeze ends: everything trains (second epoch of a fixbase_epoch = 1 run), same batch
    eng.set_model_mode("train")
    eng.two_stepped_transfer_learning(1, 1, open_layers)
    s2 = eng.forward_backward(batch())
    res["summary_step2"] = np.array([float(s2[k]) for k in KEYS])
    sd = m.state_dict()
    res["nbt_step2"] = np.array([int(sd[k]) for k in sd if k.endswith("num_batches_tracked")])
    res["step2_grad_none"] = np.array([p.grad is None for _, p in m.named_parameters()])
    for k, v in res.items():
        out[tag + "/" + k] = v
    print(tag, "loss", res["summary"][0], "-> step 2", res["summary_step2"][0], "frozen grads None:", int(res["grad_none"].sum()),
          "buffers unchanged:", int(res["buffer_unchanged"].sum()), "of", len(bn_keys), flush=True)


case("fix_cls", ["classifier_R", "classifier_N", "classifier_T"], seed=6)
case("fix_backbone", ["backbone"], seed=7)
path = os.path.join(HERE, "model_golden_r4.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path))
