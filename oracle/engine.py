"""ORACLE (test infrastructure, never imported by the product): CPU restatement of the reference's training /
evaluation loop for the hot path, on top of oracle/model.py and oracle/evaluator.py.

  evaluate   <- torchreid/engine/engine.py:339-441 (_evaluate: features per loader batch, concatenated in loader order;
                squared-Euclidean distmat; Market-1501 CMC / mAP; returns (rank-1, mAP))
  run        <- torchreid/engine/engine.py:126-232 + train :234-282 (epoch loop; scheduler stepped once per epoch
                AFTER the epoch's batches; evaluation + checkpoint when (epoch+1) >= start_eval, eval_freq > 0,
                (epoch+1) % eval_freq == 0 and (epoch+1) != max_epoch -- never after the last epoch)
Pinned against the imported reference by tests/golden/model_golden_r2.npz (evalpipe, run2) in tests/test_engine_oracle.py."""
import numpy as np
import torch

from . import evaluator as ev
from . import model as om


def features(sd, loader, **flags):
    f, pids, cams = [], [], []
    with torch.no_grad():
        for data in loader:
            f.append(om.forward(sd, data["img"], False, **flags))
            pids.extend(int(p) for p in data["pid"])
            cams.extend(int(c) for c in data["camid"])
    return torch.cat(f, 0).numpy(), np.asarray(pids), np.asarray(cams)


def evaluate(sd, query_loader, gallery_loader, **flags):
    qf, qp, qc = features(sd, query_loader, **flags)
    gf, gp, gc = features(sd, gallery_loader, **flags)
    dist = ev.sqeuclid_np(qf, gf)
    cmc, m_ap = ev.rank_market1501_c(dist, qp, gp, qc, gc)
    return cmc, m_ap, qf, gf, dist


def run(sd, train_loader, num_classes, max_epoch, start_epoch=0, start_eval=0, eval_freq=-1, lr=1e-3, milestones=(),
        gamma=0.1, test_loaders=None, margin=1.0):
    """returns (final state, per-batch summaries, [(epoch, rank1, mAP)], epochs at which a checkpoint is written, lr)"""
    mom = None
    summaries, evals, saved = [], [], []
    for epoch in range(start_epoch, max_epoch):
        for data in train_loader:
            s, _, sd, mom = om.train_step(sd, data["img"], data["pid"], num_classes, lr=lr, mom_state=mom, margin=margin)
            summaries.append(s)
        if (epoch + 1) in milestones:               # MultiStepLR, stepped once per epoch (engine.py:282)
            lr = lr * gamma
        if (epoch + 1) >= start_eval and eval_freq > 0 and (epoch + 1) % eval_freq == 0 and (epoch + 1) != max_epoch:
            cmc, m_ap = evaluate(sd, test_loaders["query"], test_loaders["gallery"])[:2]
            evals.append((epoch, float(cmc[0]), float(m_ap)))
            saved.append(epoch + 1)
    return sd, summaries, evals, saved, lr
