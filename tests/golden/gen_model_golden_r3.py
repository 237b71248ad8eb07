"""Round-3 golden vectors: the reference's evaluation pipeline (Engine.test() -> _evaluate, torchreid/engine/engine.py:
287-441) on a WELL-CONDITIONED fixture for the bf16 tests.

The round-2 `evalpipe` fixture evaluates a random-init trunk: train/eval BatchNorm over random weights is chaotic, bf16
rounding grows to a 60-80 % descriptor error for ANY bf16 implementation (stock torch bf16 autocast: 0.80) and rankings
on it cannot be compared across precisions.  Here every bottleneck's last BatchNorm scale (bn3.weight) is multiplied by
0.25 -- the residual stream dominates, as in a trained ResNet -- which brings the bf16 drift down to ~8-10 %:

  evalpipe_tame        identity images with noise 0.5: identities separate cleanly (reference mAP = 1, CMC = 1)
  evalpipe_tame_hard   noise 3.0: a non-trivial ranking (reference mAP ~ 0.92)
Each: descriptors, distmat, CMC, mAP as the reference computes them on the CPU in fp32.
Run (this container only, needs /root/reference):  python tests/golden/gen_model_golden_r3.py"""
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
from tests.util_model import G_CAMS, G_PIDS, Q_CAMS, Q_PIDS, TAME_SCALE, tame_  # noqa: E402

import_reference()
from torchreid.models import build_model  # noqa: E402
from torchreid.engine import Image3MEngine  # noqa: E402
import torchreid.engine.engine as E  # noqa: E402

torch.set_num_threads(8)
out = {}


class FakeDM:
    sources = ["synthetic"]
    train_loader = []

    def __init__(self, C, test_loader):
        self.num_train_pids = C
        self.test_loader = test_loader


def id_images(pids, cams, seed, noise):
    return [torch.from_numpy(x) for x in detgen.generate_identity_images(pids, cams, seed, noise=noise)]


def loader(n, seed, pids, cams, noise, bs=4):
    xs = id_images(pids, cams, seed, noise)
    return [{"img": [x[i:i + bs] for x in xs], "pid": torch.as_tensor(pids[i:i + bs]), "camid": torch.as_tensor(cams[i:i + bs]),
             "impath": "", "timeid": torch.zeros(len(pids[i:i + bs]))} for i in range(0, n, bs)]


def eval_case(tag, seed, noise):
    m = build_model("ieee3modalPart", num_classes=171, loss="margin", pretrained=False, use_gpu=False)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    st = {k: torch.from_numpy(np.asarray(v)) for k, v in detgen.generate_state(shapes, seed=seed).items()}
    m.load_state_dict(tame_(st))
    bns = [b for b in m.modules() if isinstance(b, torch.nn.modules.batchnorm._BatchNorm)]
    for b in bns:                       # running statistics := batch statistics of the gallery images (momentum 1)
        b.momentum = 1.0
    m.train()
    with torch.no_grad():
        m([x.clone() for x in id_images(G_PIDS, G_CAMS, 12, noise)])
    for b in bns:
        b.momentum = 0.1
        b.num_batches_tracked.zero_()
    m.eval()
    dm = FakeDM(171, {"synthetic": {"query": loader(8, 11, Q_PIDS, Q_CAMS, noise), "gallery": loader(24, 12, G_PIDS, G_CAMS, noise)}})
    eng = Image3MEngine(dm, m, torch.optim.SGD(m.parameters(), lr=1e-3), margin=1, use_gpu=False)
    grabbed = {}
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
        with redirect_stdout(io.StringIO()):
            m_ap = eng.test()
    finally:
        E.compute_distance_matrix, E.evaluate_rank = orig_cdm, orig_rank
    for k, v in grabbed.items():
        out[tag + "/" + k] = v
    out[tag + "/noise"], out[tag + "/bn3_scale"] = float(noise), float(TAME_SCALE)
    print(tag, "mAP", m_ap, "cmc", grabbed["cmc"][:5], flush=True)


eval_case("evalpipe_tame", seed=8, noise=0.5)
eval_case("evalpipe_tame_hard", seed=8, noise=3.0)
path = os.path.join(HERE, "model_golden_r3.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path))
