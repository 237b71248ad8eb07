"""Generates tests/golden/evaluator_golden.npz by running the IMPORTED REFERENCE
(torchreid.metrics.distance / torchreid.metrics.rank) in this container.
Run:  python tests/golden/gen_evaluator_golden.py
Cases: the reference's own test_cython.py shape (30x300,
max_rank 5, rank_cylib/test_cython.py:28-35), 64x512 at d=768 and d=2304 with
integer-grid features (exact fp32 dot products), a query with no valid gallery
match, num_g < max_rank, and a float-feature distmat for tolerance checks.
Distance matrices fed to evaluate_rank are made tie-free (see `detie`) because
the reference's argsort is unstable on exact ties."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.ref_import import import_reference  # noqa: E402

import_reference()
from torchreid.metrics.distance import compute_distance_matrix  # noqa: E402
from torchreid.metrics.rank import evaluate_rank  # noqa: E402


def detie(d):
    """integer-valued distmat + j/1024: exactly representable, tie-free, order = (dist, index)."""
    assert np.all(d == np.round(d)) and d.max() < 8192 and d.shape[1] <= 1024
    return (d + (np.arange(d.shape[1], dtype=np.float32) / 1024.0)[None, :]).astype(np.float32)


out = {}
rng = np.random.RandomState(1234)

# case A: the reference's own timing-script shape
nq, ng = 30, 300
dA = (rng.rand(nq, ng) * 20).astype(np.float32)
qp = rng.randint(0, nq, size=nq); gp = rng.randint(0, nq, size=ng)
qc = rng.randint(0, 5, size=nq); gc = rng.randint(0, 5, size=ng)
cmc, mAP = evaluate_rank(dA, qp, gp, qc, gc, max_rank=5, use_cython=False)
out.update(A_dist=dA, A_qp=qp, A_gp=gp, A_qc=qc, A_gc=gc, A_cmc=cmc, A_map=np.float64(mAP), A_maxrank=5)

# cases B/C: integer-grid features, d=768 / 2304
for tag, d in (("B", 768), ("C", 2304)):
    nq, ng = 64, 512
    qf = rng.randint(0, 3, size=(nq, d)).astype(np.float32)
    gf = rng.randint(0, 3, size=(ng, d)).astype(np.float32)
    dm = compute_distance_matrix(torch.from_numpy(qf), torch.from_numpy(gf), "euclidean").numpy()
    assert np.all(dm == np.round(dm))
    qp = rng.randint(0, 40, size=nq); gp = rng.randint(0, 40, size=ng)
    qc = rng.randint(0, 4, size=nq); gc = rng.randint(0, 4, size=ng)
    dt = detie(dm)
    cmc, mAP = evaluate_rank(dt, qp, gp, qc, gc)   # engine passes no max_rank -> 20
    out.update({tag + "_qf": qf.astype(np.int8), tag + "_gf": gf.astype(np.int8), tag + "_dist": dm,
                tag + "_qp": qp, tag + "_gp": gp, tag + "_qc": qc, tag + "_gc": gc,
                tag + "_cmc": cmc, tag + "_map": np.float64(mAP),
                tag + "_argsort0": np.argsort(dt[0], kind="stable").astype(np.int32)})

# case D: one query whose identity is absent from the gallery (skipped, rank.py:142-144)
nq, ng = 8, 64
dD = detie(rng.randint(0, 500, size=(nq, ng)).astype(np.float32))
qp = np.arange(nq); qp[3] = 999
gp = rng.randint(0, nq, size=ng); qc = np.zeros(nq, np.int64); gc = rng.randint(0, 3, size=ng)
cmc, mAP = evaluate_rank(dD, qp, gp, qc, gc)
out.update(D_dist=dD, D_qp=qp, D_gp=gp, D_qc=qc, D_gc=gc, D_cmc=cmc, D_map=np.float64(mAP))

# case E: num_g < max_rank (rank.py:110-115 clamps and prints a note)
nq, ng = 6, 12
dE = detie(rng.randint(0, 500, size=(nq, ng)).astype(np.float32))
qp = rng.randint(0, 4, size=nq); gp = rng.randint(0, 4, size=ng)
# no gallery entry is removed here: with removals the reference builds a ragged list at
# rank.py:150 (cmc[:max_rank] of unequal lengths) and np.asarray(...) raises.
qc = np.zeros(nq, np.int64); gc = np.ones(ng, np.int64)
cmc, mAP = evaluate_rank(dE, qp, gp, qc, gc)
out.update(E_dist=dE, E_qp=qp, E_gp=gp, E_qc=qc, E_gc=gc, E_cmc=cmc, E_map=np.float64(mAP))

# case F: float features (post-ReLU-like |N(0,1)|), distmat for tolerance parity; cosine too
nq, ng, d = 48, 200, 768
qf = np.abs(rng.randn(nq, d)).astype(np.float32); gf = np.abs(rng.randn(ng, d)).astype(np.float32)
dm = compute_distance_matrix(torch.from_numpy(qf), torch.from_numpy(gf), "euclidean").numpy()
dc = compute_distance_matrix(torch.from_numpy(qf), torch.from_numpy(gf), "cosine").numpy()
out.update(F_qf=qf.astype(np.float16), F_gf=gf.astype(np.float16))
# features stored as fp16 to keep the fixture small: recompute on the rounded values
qf = qf.astype(np.float16).astype(np.float32); gf = gf.astype(np.float16).astype(np.float32)
dm = compute_distance_matrix(torch.from_numpy(qf), torch.from_numpy(gf), "euclidean").numpy()
dc = compute_distance_matrix(torch.from_numpy(qf), torch.from_numpy(gf), "cosine").numpy()
out.update(F_dist=dm, F_cos=dc)

# case G: all queries invalid -> AssertionError (rank.py:165)
try:
    evaluate_rank(dE, qp + 100, gp, qc, gc)
    raised = False
except AssertionError as e:
    raised = True
    out["G_msg"] = np.array(str(e))
assert raised

np.savez_compressed(os.path.join(HERE, "evaluator_golden.npz"), **out)
print("wrote", os.path.join(HERE, "evaluator_golden.npz"), os.path.getsize(os.path.join(HERE, "evaluator_golden.npz")))
