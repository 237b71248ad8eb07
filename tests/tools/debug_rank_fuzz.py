"""one-off differential fuzz of ieee_rank_market1501 (both kernel paths) against the oracle's C restatement:
random sizes, identity / camera counts, tie densities, row strides.  Run by hand on a GPU box:
    python tests/tools/debug_rank_fuzz.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import evaluator as ev  # noqa: E402
from ieee_amd.metrics import evaluate_rank  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(cases):
    nq = int(rng.randint(1, 80))
    ng = int(rng.choice([1, 2, 3, 7, 19, 64, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, rng.randint(1, 9000)]))
    nid = int(rng.choice([1, 2, 5, 50, 500]))
    ncam = int(rng.choice([1, 2, 6]))
    mode = rng.randint(0, 4)
    if mode == 0:
        d = (rng.rand(nq, ng) * 10).astype(np.float32)
    elif mode == 1:
        d = rng.randint(0, 4, size=(nq, ng)).astype(np.float32)          # dense ties
    elif mode == 2:
        d = (rng.randn(nq, ng) * 1e-3 + 5).astype(np.float32)            # narrow range
    else:
        d = (rng.rand(nq, ng) * np.exp(rng.randn(nq, 1) * 5)).astype(np.float32)   # row scales differ
    qp, gp = rng.randint(0, nid, nq), rng.randint(0, nid, ng)
    qc, gc = rng.randint(0, ncam, nq), rng.randint(0, ncam, ng)
    try:
        cmc_o, map_o = ev.rank_market1501_c(d, qp, gp, qc, gc, 20)
        ok_o = True
    except AssertionError:
        ok_o = False
    pad = int(rng.choice([0, 0, 1, 3]))
    dd = torch.zeros(nq, ng + pad, device="cuda")
    dd[:, :ng] = torch.from_numpy(d).cuda()
    dd = dd[:, :ng]
    for general in ("0", "1"):
        os.environ["IEEE_RANK_GENERAL"] = general
        try:
            cmc, m_ap = evaluate_rank(dd, qp, gp, qc, gc)
            ok = True
        except AssertionError:
            ok = False
        if ok != ok_o or (ok and (not np.array_equal(cmc, cmc_o) or abs(m_ap - map_o) > 1e-12)):
            bad += 1
            print("MISMATCH case %d general=%s nq=%d ng=%d nid=%d ncam=%d mode=%d pad=%d" % (it, general, nq, ng, nid, ncam, mode, pad))
print("%d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
