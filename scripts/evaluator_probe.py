"""config-4 evaluator under the profiler: the fp32 distmat, the bf16-input distmat, the f16x2 split path and the ranking
kernels, a few launches each (rocprofv3 --pmc passes: scripts/pmc_passes_eval.sh)"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from ieee_amd.metrics import compute_distance_matrix, evaluate_rank  # noqa: E402

g = torch.Generator(device="cpu").manual_seed(1)
Q, G, D = 10000, 100000, 768
qf = torch.randn(Q, D, generator=g).abs().cuda()
gf = torch.randn(G, D, generator=g).abs().cuda()
rs = np.random.RandomState(1)
qp, gp, qc, gc = rs.randint(0, 1000, Q), rs.randint(0, 1000, G), rs.randint(0, 4, Q), rs.randint(0, 4, G)
for _ in range(3):
    dm = compute_distance_matrix(qf, gf)
for _ in range(3):
    compute_distance_matrix(qf.bfloat16(), gf.bfloat16())
for _ in range(3):
    compute_distance_matrix(qf, gf, precision="f16x2")
for _ in range(3):
    cmc, m_ap = evaluate_rank(dm, qp, gp, qc, gc)
torch.cuda.synchronize()
print("mAP", m_ap)
