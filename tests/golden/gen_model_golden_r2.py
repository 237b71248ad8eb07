"""Generates tests/golden/model_golden_r2.npz by running the IMPORTED REFERENCE on the CPU in this
container (fp32), with weights and images from ieee_amd.detgen (regenerated identically on the GPU box).
Round-2 additions to model_golden.npz (which stays as it is):

  softmax8            MultiModalImageSoftmaxEngine.forward_backward (engine/image/softmax.py:81-132), C = 171, B = 8
  c750_train8[...]    Image3MEngine step with the 750 classes of Market1501-multimodal (BASELINE config 5), full model
                      and each ablation flag; c750_softmax8 = the 3M-off leg of that sweep
  evalpipe            Engine.test() -> _evaluate (engine/engine.py:287-417) on tiny query / gallery loaders:
                      features, distmat, CMC, mAP as the reference computes them on the CPU
  run2                Engine.run(max_epoch=2, eval_freq=1) on a synthetic datamanager (config 1's "1 epoch plumbing" +
                      the in-loop evaluation and checkpoint): per-batch summaries, the evaluation after epoch 1,
                      parameter checksums at the end
Run:  python tests/golden/gen_model_golden_r2.py"""
import io
import os
import sys
import tempfile
from contextlib import redirect_stdout

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.ref_import import import_reference  # noqa: E402
from ieee_amd import detgen  # noqa: E402

import_reference()
from torchreid.models import build_model  # noqa: E402
from torchreid.engine import Image3MEngine, MultiModalImageSoftmaxEngine  # noqa: E402

out = {}
torch.set_num_threads(8)


class FakeDM:
    train_loader = []
    test_loader = {}
    sources = ["synthetic"]

    def __init__(self, C, train_loader=None, test_loader=None):
        self.num_train_pids = C
        self.train_loader = train_loader or []
        self.test_loader = test_loader or {}


def make_model(seed, C, loss):
    m = build_model("ieee3modalPart", num_classes=C, loss=loss, pretrained=False, use_gpu=False)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    st = detgen.generate_state(shapes, seed=seed)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
    return m


def stats(t):
    t = t.detach().double().flatten()
    idx = torch.linspace(0, t.numel() - 1, 32).long()
    return np.concatenate([[t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()], t[idx].numpy()])


def images(B, seed):
    return [torch.from_numpy(x) for x in detgen.generate_images(B, seed=seed)]


def sgd(m):
    return torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9, weight_decay=5e-4, dampening=0, nesterov=True)


def record_state(tag, m, full):
    names = [k for k, _ in m.named_parameters()]
    out[tag + "/grad_none"] = np.array([p.grad is None for _, p in m.named_parameters()])
    if not full:
        return
    out[tag + "/param_names"] = np.array(names)
    out[tag + "/grad_stats"] = np.stack([stats(p.grad) if p.grad is not None else np.zeros(35) for _, p in m.named_parameters()])
    sd = m.state_dict()
    out[tag + "/post_param_stats"] = np.stack([stats(sd[k]) for k in names])
    bn_keys = [k for k in sd if k.endswith("running_mean") or k.endswith("running_var")]
    out[tag + "/buffer_names"] = np.array(bn_keys)
    out[tag + "/post_buffer_stats"] = np.stack([stats(sd[k]) for k in bn_keys])


def run_step(tag, engine_cls, loss, C, B, seed, flags=None, full=False):
    m = make_model(seed, C, loss)
    for k, v in (flags or {}).items():
        setattr(m, k, v)
    pids = torch.arange(B) // 4
    if engine_cls is Image3MEngine:
        eng = Image3MEngine(FakeDM(C), m, sgd(m), margin=1, weight_m=1, weight_x=1, use_gpu=False, label_smooth=True)
        keys = ("loss", "LossX", "LossM", "lossR", "lossN", "lossT", "accR", "accN", "accT")
    else:
        eng = MultiModalImageSoftmaxEngine(FakeDM(C), m, sgd(m), use_gpu=False, label_smooth=True)
        keys = ("loss_all", "loss_R", "acc_R", "loss_N", "acc_N", "loss_T", "acc_T")
    m.train()
    captured = {}
    orig = m.forward

    def fwd(*a, **k):
        o = orig(*a, **k)
        captured["out"] = o
        return o
    m.forward = fwd
    s = eng.forward_backward({"img": images(B, seed), "pid": pids, "camid": pids * 0, "impath": "", "timeid": pids * 0})
    m.forward = orig
    o = captured["out"]
    out[tag + "/logits"] = torch.stack([torch.stack(list(x)) for x in o[:3]]).detach().numpy().reshape(18, B, C)
    if len(o) == 6:
        out[tag + "/feats"] = torch.stack(list(o[3:])).detach().numpy()
    out[tag + "/summary"] = np.array([float(s[k]) for k in keys])
    out[tag + "/summary_keys"] = np.array(keys)
    record_state(tag, m, full)
    print(tag, dict(zip(keys, out[tag + "/summary"])), flush=True)


# ---- CE-only engine (A22) and the 750-class configuration (config 5)
run_step("softmax8", MultiModalImageSoftmaxEngine, "softmax", 171, 8, seed=6, full=True)
run_step("c750_train8", Image3MEngine, "margin", 750, 8, seed=7, full=True)
run_step("c750_softmax8", MultiModalImageSoftmaxEngine, "softmax", 750, 8, seed=7)
run_step("c750_train8_noatt", Image3MEngine, "margin", 750, 8, seed=7, flags={"attention": False})
run_step("c750_train8_nocim", Image3MEngine, "margin", 750, 8, seed=7, flags={"interaction": False})
run_step("c750_train8_norem", Image3MEngine, "margin", 750, 8, seed=7, flags={"using_REM": False})


# ---- evaluation pipeline (A17 / N1): tiny loaders in the reference's batch-dict format
def id_images(pids, cams, seed):
    return [torch.from_numpy(x) for x in detgen.generate_identity_images(pids, cams, seed, noise=0.5)]


def calibrate(m, imgs):
    """running statistics := the batch statistics of `imgs` (one train-mode forward with momentum 1): a random-init
    trunk under the GENERATED running statistics maps every image to nearly the same descriptor and the distances are
    rounding noise; calibrated, descriptors separate by identity and the ranking is well conditioned.  The tests do the
    same with the oracle (oracle.model.MOM = 1)."""
    bns = [b for b in m.modules() if isinstance(b, torch.nn.modules.batchnorm._BatchNorm)]
    for b in bns:
        b.momentum = 1.0
    m.train()
    with torch.no_grad():
        m([x.clone() for x in imgs])
    for b in bns:
        b.momentum = 0.1
        b.num_batches_tracked.zero_()
    m.eval()


def loader(n, seed, pids, cams, bs=4):
    xs = id_images(pids, cams, seed)
    return [{"img": [x[i:i + bs] for x in xs], "pid": torch.as_tensor(pids[i:i + bs]), "camid": torch.as_tensor(cams[i:i + bs]),
             "impath": "", "timeid": torch.zeros(len(pids[i:i + bs]))} for i in range(0, n, bs)]


Q_PIDS = [0, 1, 2, 3, 0, 1, 2, 3]
Q_CAMS = [0] * 8
G_PIDS = [0, 0, 1, 1, 2, 2, 3, 3, 0, 1, 2, 3, 4, 4, 5, 5, 0, 1, 2, 3, 6, 6, 7, 7]
G_CAMS = [1, 2, 1, 2, 1, 2, 1, 2, 0, 0, 0, 0, 1, 2, 1, 2, 3, 3, 3, 3, 1, 2, 1, 2]      # cam-0 entries: removed for a same-pid query


def eval_case(tag, seed):
    m = make_model(seed, 171, "margin")
    calibrate(m, id_images(G_PIDS, G_CAMS, 12))
    dm = FakeDM(171, test_loader={"synthetic": {"query": loader(8, 11, Q_PIDS, Q_CAMS), "gallery": loader(24, 12, G_PIDS, G_CAMS)}})
    eng = Image3MEngine(dm, m, sgd(m), margin=1, use_gpu=False)
    grabbed = {}
    import torchreid.engine.engine as E
    orig_cdm, orig_rank = E.compute_distance_matrix, E.evaluate_rank

    def cdm(a, b, metric):
        grabbed["qf"], grabbed["gf"] = a.numpy().copy(), b.numpy().copy()
        d = orig_cdm(a, b, metric)
        grabbed["distmat"] = d.numpy().copy()
        return d

    def er(*a, **k):
        cmc, m_ap = orig_rank(*a, **k)
        grabbed["cmc"], grabbed["mAP"] = np.asarray(cmc).copy(), float(m_ap)
        return cmc, m_ap
    E.compute_distance_matrix, E.evaluate_rank = cdm, er
    try:
        with redirect_stdout(io.StringIO()) as buf:
            m_ap = eng.test()
    finally:
        E.compute_distance_matrix, E.evaluate_rank = orig_cdm, orig_rank
    for k, v in grabbed.items():
        out[tag + "/" + k] = v
    out[tag + "/returned_mAP"] = float(m_ap)
    out[tag + "/printed"] = np.array(buf.getvalue())
    out[tag + "/q_pids"], out[tag + "/q_cams"] = np.array(Q_PIDS), np.array(Q_CAMS)
    out[tag + "/g_pids"], out[tag + "/g_cams"] = np.array(G_PIDS), np.array(G_CAMS)
    out[tag + "/bn_check"] = stats(torch.cat([v.flatten() for k, v in m.state_dict().items() if "running_" in k]))
    d = np.sort(grabbed["distmat"], 1)
    print(tag, "mAP", m_ap, "cmc", grabbed["cmc"][:5], "min gap between neighbours in a sorted row", np.diff(d, axis=1).min(), flush=True)


eval_case("evalpipe", seed=8)


# ---- Engine.run: 2 epochs x 2 batches of B = 4 (config 1 shape), evaluation + checkpoint after epoch 1
def run_case(tag, seed):
    m = make_model(seed, 171, "margin")
    calibrate(m, id_images(G_PIDS, G_CAMS, 12))
    train = []
    for i in range(2):
        pids = torch.full((4,), i, dtype=torch.long)
        train.append({"img": id_images([i] * 4, [0, 1, 2, 3], 20 + i), "pid": pids, "camid": pids * 0, "impath": "",
                      "timeid": pids * 0})
    dm = FakeDM(171, train_loader=train,
                test_loader={"synthetic": {"query": loader(8, 11, Q_PIDS, Q_CAMS), "gallery": loader(24, 12, G_PIDS, G_CAMS)}})
    opt = sgd(m)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[1], gamma=0.1)
    eng = Image3MEngine(dm, m, opt, margin=1, scheduler=sched, use_gpu=False)
    summaries, evals = [], []
    fb = eng.forward_backward

    def fb_rec(data):
        data = dict(data)
        data["img"] = [x.clone() for x in data["img"]]
        s = fb(data)
        summaries.append([float(s[k]) for k in ("loss", "LossX", "LossM", "lossR", "lossN", "lossT", "accR", "accN", "accT")])
        return s
    eng.forward_backward = fb_rec
    ev = eng._evaluate

    def ev_rec(**k):
        r1, m_ap = ev(**k)
        evals.append([float(r1), float(m_ap)])
        return r1, m_ap
    eng._evaluate = ev_rec
    with tempfile.TemporaryDirectory() as d, redirect_stdout(io.StringIO()) as buf:
        eng.run(save_dir=d, max_epoch=2, eval_freq=1, print_freq=1)
        saved = sorted(os.listdir(os.path.join(d, "model"))) if os.path.isdir(os.path.join(d, "model")) else []
    out[tag + "/summaries"] = np.array(summaries)
    out[tag + "/evals"] = np.array(evals)
    out[tag + "/saved"] = np.array(saved)
    out[tag + "/final_lr"] = float(opt.param_groups[0]["lr"])
    out[tag + "/printed"] = np.array(buf.getvalue())
    sd = m.state_dict()
    names = [k for k, _ in m.named_parameters()]
    out[tag + "/param_names"] = np.array(names)
    out[tag + "/final_param_stats"] = np.stack([stats(sd[k]) for k in names])
    out[tag + "/nbt"] = np.array([int(sd[k]) for k in sd if k.endswith("num_batches_tracked")])
    print(tag, "summaries", np.array(summaries)[:, 0], "evals", evals, "saved", saved, flush=True)


run_case("run2", seed=9)

path = os.path.join(HERE, "model_golden_r2.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path))
