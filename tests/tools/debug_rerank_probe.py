"""timing of the device re-ranking (N3) and of the oracle on a small sample"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from ieee_amd.metrics.distance import compute_distance_matrix  # noqa: E402
from ieee_amd.rerank import re_ranking  # noqa: E402

for Q, G in ((836, 836), (3000, 12000)):
    rng = np.random.RandomState(0)
    cen = rng.randn(200, 64) * 2
    qf = torch.from_numpy((cen[rng.randint(0, 200, Q)] + rng.randn(Q, 64)).astype(np.float32)).cuda()
    gf = torch.from_numpy((cen[rng.randint(0, 200, G)] + rng.randn(G, 64)).astype(np.float32)).cuda()
    qg, qq, gg = compute_distance_matrix(qf, gf), compute_distance_matrix(qf, qf), compute_distance_matrix(gf, gf)
    re_ranking(qg, qq, gg)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        out = re_ranking(qg, qq, gg)
    torch.cuda.synchronize()
    print("Q=%d G=%d: %.1f ms on the device" % (Q, G, (time.time() - t0) / 3 * 1e3))
    if Q < 1000:
        from oracle import rerank as orr
        t0 = time.time()
        want = orr.re_ranking(qg.cpu().numpy(), qq.cpu().numpy(), gg.cpu().numpy())
        print("   oracle (numpy, the reference's algorithm): %.2f s; max |diff| %.2e" % (time.time() - t0, np.abs(out.cpu().numpy() - want).max()))
