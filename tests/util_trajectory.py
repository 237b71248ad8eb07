"""Shared by tests/test_trajectory_gpu.py and scripts/trajectory_probe.py: a short REAL training run of the native engines
(reference loop: torchreid/engine/engine.py:126-283, step: engine/image/margin.py:94-154, optimizer / schedule:
optim/optimizer.py:130-138, optim/lr_scheduler.py:46-60, recipe: configs/RGBNT_ieee_part_margin.yaml:17-38) on
identity-separable synthetic triples (ieee_amd/detgen.py: generate_identity_images), in either arithmetic mode."""
import contextlib
import io

import numpy as np
import torch

from ieee_amd import detgen
from tests.util_model import generated_state, tame_


def make_train_set(n_ids, per_id, seed, noise):
    """n_ids x per_id triples; identity i owns rows [i * per_id, (i + 1) * per_id); camera = instance index"""
    pids = [i for i in range(n_ids) for _ in range(per_id)]
    cams = [j for _ in range(n_ids) for j in range(per_id)]
    xs = [torch.from_numpy(x) for x in detgen.generate_identity_images(pids, cams, seed, noise=noise)]
    return xs, torch.as_tensor(pids), torch.as_tensor(cams)


def epoch_batches(xs, pids, cams, n_ids, per_id, ids_per_batch, k, batches, rs):
    """what RandomIdentitySampler hands the engine (data/sampler.py:27-90): batches of ids_per_batch identities x k
    instances, identity-contiguous (the 3M loss chunks the batch by identity, multi_modal_margin_loss_new.py:24-33)"""
    out = []
    for _ in range(batches):
        ids = rs.choice(n_ids, ids_per_batch, replace=False)
        rows = np.concatenate([i * per_id + rs.choice(per_id, k, replace=False) for i in ids])
        rows = torch.as_tensor(rows)
        out.append({"img": [x[rows] for x in xs], "pid": pids[rows], "camid": cams[rows], "impath": "",
                    "timeid": torch.zeros(len(rows))})
    return out


def eval_loaders(n_ids, seed, noise, bs=16, q_per_id=6, g_per_id=8):
    """query: q_per_id triples per identity from camera 0; gallery: g_per_id per identity from cameras 1..4 (the evaluator
    drops same-identity-same-camera pairs, so every query has g_per_id true matches)"""
    def loader(pids, cams, sd):
        xs = [torch.from_numpy(x) for x in detgen.generate_identity_images(pids, cams, sd, noise=noise)]
        return [{"img": [x[i:i + bs] for x in xs], "pid": torch.as_tensor(pids[i:i + bs]),
                 "camid": torch.as_tensor(cams[i:i + bs]), "impath": "", "timeid": torch.zeros(len(pids[i:i + bs]))}
                for i in range(0, len(pids), bs)]
    q_p = [i for i in range(n_ids) for _ in range(q_per_id)]
    q_c = [0] * len(q_p)
    g_p = [i for i in range(n_ids) for _ in range(g_per_id)]
    g_c = [1 + j % 4 for _ in range(n_ids) for j in range(g_per_id)]
    return {"query": loader(q_p, q_c, seed + 1), "gallery": loader(g_p, g_c, seed + 2)}


class _DM(object):
    def __init__(self, C, train_loader, test_loader):
        self.num_train_pids = C
        self.train_loader = train_loader
        self.test_loader = test_loader
        self.sources = ["synthetic"]
        self.num_instances = 4


def run_training(dtype, state, C=171, engine="margin", flags=None, n_ids=8, per_id=8, ids_per_batch=4, k=4,
                 batches_per_epoch=10, epochs=12, milestones=(8, 10), lr=1e-3, data_seed=5, noise=0.5, eval_noise=0.5,
                 perturb=0.0, train_set=None, test_loader=None, after_step=None):
    """Trains a fresh model from `state` (a reference-keyed state dict) for epochs x batches_per_epoch engine steps through
    Engine.run (SGD-nesterov momentum 0.9, weight decay 5e-4, MultiStepLR gamma 0.1 at `milestones`, label smoothing,
    margin 1) and evaluates with Engine.test().  perturb: relative jitter of the starting parameters (the fp32-vs-fp32
    control).  Returns per-step losses / accuracies, the mAP / CMC of the final evaluation and the model."""
    from ieee_amd.engine import Image3MEngine, MultiModalImageSoftmaxEngine
    from ieee_amd.models import build_model
    from ieee_amd.optim import build_lr_scheduler, build_optimizer
    flags = dict(flags or {})
    m = build_model("ieee3modalPart", num_classes=C, loss="margin" if engine == "margin" else "softmax", pretrained=False,
                    compute_dtype=dtype, **flags)
    m.load_state_dict(state)
    if perturb:
        with torch.no_grad():
            u = torch.rand(m._flat_params.shape, generator=torch.Generator(device="cuda").manual_seed(7), device="cuda") * 2 - 1
            m._flat_params.mul_(1 + u * perturb)
    opt = build_optimizer(m, optim="sgd", lr=lr, weight_decay=5e-4, momentum=0.9)
    sched = build_lr_scheduler(opt, "multi_step", stepsize=list(milestones), gamma=0.1)
    xs, pids, cams = train_set if train_set is not None else make_train_set(n_ids, per_id, data_seed, noise)
    rs = np.random.RandomState(data_seed)
    all_batches = [epoch_batches(xs, pids, cams, n_ids, per_id, ids_per_batch, k, batches_per_epoch, rs) for _ in range(epochs)]
    if test_loader is None:     # eval_noise: one noise level or {name: level} for several evaluation sets
        levels = eval_noise if isinstance(eval_noise, dict) else {"synthetic": eval_noise}
        test_loader = {name: eval_loaders(n_ids, data_seed + 100, lv) for name, lv in levels.items()}
    dm = _DM(C, all_batches[0], test_loader)
    if engine == "margin":
        eng = Image3MEngine(dm, m, opt, margin=1, weight_m=1, weight_x=1, scheduler=sched, use_gpu=True, label_smooth=True)
        loss_key, acc_keys = "loss", ("accR", "accN", "accT")
    else:
        eng = MultiModalImageSoftmaxEngine(dm, m, opt, scheduler=sched, use_gpu=True, label_smooth=True)
        loss_key, acc_keys = "loss_all", ("acc_R", "acc_N", "acc_T")
    seen = []
    step = eng.forward_backward

    def logged(data):
        s = step(data)
        seen.append(s)
        if after_step is not None:
            after_step(len(seen) - 1, eng, m)
        return s
    eng.forward_backward = logged
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        for e in range(epochs):            # Engine.run's loop body (engine.py:190-214) with this epoch's batches
            eng.train_loader = all_batches[e]
            eng.epoch, eng.max_epoch = e, epochs
            eng.train(print_freq=1000)
        cmc_map, evals = [], {}
        orig = eng._evaluate

        def keep(**kw):
            r = orig(**kw)
            cmc_map.append(r)
            evals[kw.get("dataset_name", "")] = (float(r[0]), float(r[1]))
            return r
        eng._evaluate = keep
        eng.test()
    losses = np.array([float(s[loss_key]) for s in seen])
    accs = np.array([np.mean([float(s[a]) for a in acc_keys]) for s in seen])
    extra = {}
    if engine == "margin":
        extra["LossM"] = np.array([float(s["LossM"]) for s in seen])
        extra["LossX"] = np.array([float(s["LossX"]) for s in seen])
    return dict(loss=losses, acc=accs, rank1=float(cmc_map[-1][0]), mAP=float(cmc_map[-1][1]), evals=evals, model=m, engine=eng,
                lr_end=opt.param_groups[0]["lr"], report=sink.getvalue(), **extra)


def tamed_state(C, seed=11):
    from ieee_amd._spec import state_spec
    shapes = {k: s for k, s, _ in state_spec(C)}
    return tame_(generated_state(shapes, seed))


def smooth(x, w=5):
    return np.convolve(x, np.ones(w) / w, mode="valid")
