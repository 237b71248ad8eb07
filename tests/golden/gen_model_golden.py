"""Generates tests/golden/model_golden.npz by running the IMPORTED REFERENCE on the CPU in this
container (build_model('ieee3modalPart') + Image3MEngine.forward_backward, fp32), with weights and
images from ieee_amd.detgen (regenerated identically on the GPU box).  Also records the reference's
own noise floor (1 vs 8 CPU threads) next to the train-mode tensors.
Run:  python tests/golden/gen_model_golden.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.ref_import import import_reference  # noqa: E402
from ieee_amd import detgen  # noqa: E402

import_reference()
from torchreid.models import build_model  # noqa: E402
from torchreid.engine import Image3MEngine  # noqa: E402

C = 171
out = {}


class FakeDM:
    num_train_pids = C
    train_loader = []
    test_loader = {}
    sources = ["synthetic"]


def make_model(seed):
    m = build_model("ieee3modalPart", num_classes=C, loss="margin", pretrained=False, use_gpu=False)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    st = detgen.generate_state(shapes, seed=seed)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
    return m


def stats(t):
    t = t.detach().double().flatten()
    idx = torch.linspace(0, t.numel() - 1, 32).long()
    return np.concatenate([[t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()], t[idx].numpy()])


def run_step(B, K, seed, threads, tag, flags=None, full=True):
    torch.set_num_threads(threads)
    m = make_model(seed)
    for k, v in (flags or {}).items():
        setattr(m, k, v)
    xs = [torch.from_numpy(x) for x in detgen.generate_images(B, seed=seed)]
    pids = torch.arange(B) // K
    opt = torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9, weight_decay=5e-4, dampening=0, nesterov=True)
    eng = Image3MEngine(FakeDM(), m, opt, margin=1, weight_m=1, weight_x=1, use_gpu=False, label_smooth=True)
    m.train()
    # outputs of the forward inside the step (hook the model call)
    captured = {}
    orig = m.forward

    def fwd(*a, **k):
        o = orig(*a, **k)
        captured["out"] = o
        return o
    m.forward = fwd
    summary = eng.forward_backward({"img": [x.clone() for x in xs], "pid": pids, "camid": pids * 0, "impath": "",
                                    "timeid": pids * 0})
    m.forward = orig
    oR, oN, oT, fR, fN, fT = captured["out"]
    res = {}
    res["logits"] = torch.stack([torch.stack(list(o)) for o in (oR, oN, oT)]).detach().numpy().reshape(18, B, C)
    res["feats"] = torch.stack([fR, fN, fT]).detach().numpy()
    res["summary"] = np.array([float(summary[k]) for k in
                               ("loss", "LossX", "LossM", "lossR", "lossN", "lossT", "accR", "accN", "accT")])
    if full:
        names = [k for k, _ in m.named_parameters()]
        gs = []
        none_mask = []
        for k, p in m.named_parameters():
            none_mask.append(p.grad is None)
            gs.append(stats(p.grad) if p.grad is not None else np.zeros(35))
        res["grad_stats"] = np.stack(gs)
        res["grad_none"] = np.array(none_mask)
        res["param_names"] = np.array(names)
        sd = m.state_dict()
        res["post_param_stats"] = np.stack([stats(sd[k]) for k in names])
        bn_keys = [k for k in sd if k.endswith("running_mean") or k.endswith("running_var")]
        res["buffer_names"] = np.array(bn_keys)
        res["post_buffer_stats"] = np.stack([stats(sd[k]) for k in bn_keys])
        res["nbt"] = np.array([int(sd[k]) for k in sd if k.endswith("num_batches_tracked")])
        for k in ("classifier_R.0.weight", "REM.0.param", "REM.1.conv_part.bias", "backbone.0.bn1.weight",
                  "reduce_layer.2.layers.1.weight", "fc_T.3.1.bias", "backbone.1.conv1.weight"):
            res["grad:" + k] = dict(m.named_parameters())[k].grad.numpy().copy()
        res["post:reduce_layer.0.layers.1.running_var"] = sd["reduce_layer.0.layers.1.running_var"].numpy().copy()
        res["post:backbone.2.layer4.2.bn3.running_mean"] = sd["backbone.2.layer4.2.bn3.running_mean"].numpy().copy()
    for k, v in res.items():
        out[tag + "/" + k] = v
    return m, xs, res


# ---- eval-mode golden (strict 1e-3 end to end): B=4
torch.set_num_threads(8)
m = make_model(seed=1)
m.eval()
xs = [torch.from_numpy(x) for x in detgen.generate_images(4, seed=1)]
with torch.no_grad():
    out["eval/fc_all"] = m(xs, torch.zeros(4)).numpy()        # 2nd arg = junk timeids, as engine.py:366 passes
for fl, tag in (({"attention": False}, "eval_noatt"), ({"interaction": False}, "eval_nocim"),
                ({"using_REM": False}, "eval_norem")):
    for k, v in fl.items():
        setattr(m, k, v)
    with torch.no_grad():
        out[tag + "/fc_all"] = m(xs).numpy()
    for k in fl:
        setattr(m, k, True)

# ---- train-mode goldens
_, _, r8 = run_step(16, 4, seed=2, threads=8, tag="train16")
_, _, r1 = run_step(16, 4, seed=2, threads=1, tag="train16_t1", full=False)
out["train16/noise_logits"] = np.abs(r8["logits"] - r1["logits"]).max()
out["train16/noise_feats"] = np.abs(r8["feats"] - r1["feats"]).max()
run_step(8, 4, seed=3, threads=8, tag="train8")                      # 2 identities: pins chunk()/3M
run_step(4, 4, seed=4, threads=8, tag="train4", full=False)         # BASELINE config C1 shape (1 identity)
run_step(8, 4, seed=5, threads=8, tag="train8_noatt", flags={"attention": False}, full=False)
run_step(8, 4, seed=5, threads=8, tag="train8_nocim", flags={"interaction": False}, full=False)
run_step(8, 4, seed=5, threads=8, tag="train8_norem", flags={"using_REM": False}, full=False)
print("noise floor (1 vs 8 threads): logits %.3e feats %.3e" % (out["train16/noise_logits"], out["train16/noise_feats"]))
path = os.path.join(HERE, "model_golden.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path))
