"""Generates tests/golden/rerank_golden.npz by running the REFERENCE's torchreid/utils/rerank.py::re_ranking
(imported from /root/reference in this container) on small seeded feature sets.  Run:
    python tests/golden/gen_rerank_golden.py"""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle.ref_import import import_reference  # noqa: E402


def sq_dist(a, b):
    return ((a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2.0 * (a @ b.T)).astype(np.float32)


def main():
    import_reference()
    from torchreid.utils.rerank import re_ranking
    rng = np.random.RandomState(11)
    out = {}
    for case, (Q, G, D, k1, k2, lam) in enumerate([(24, 70, 16, 20, 6, 0.3), (9, 40, 8, 6, 1, 0.5), (31, 101, 12, 10, 3, 0.2)]):
        centers = rng.randn(7, D) * 2.0
        qf = (centers[rng.randint(0, 7, Q)] + rng.randn(Q, D)).astype(np.float64)
        gf = (centers[rng.randint(0, 7, G)] + rng.randn(G, D)).astype(np.float64)
        qg, qq, gg = sq_dist(qf, gf), sq_dist(qf, qf), sq_dist(gf, gf)
        qq = np.maximum(qq, 0); gg = np.maximum(gg, 0)
        np.fill_diagonal(qq, 0.0); np.fill_diagonal(gg, 0.0)
        final = re_ranking(qg, qq, gg, k1=k1, k2=k2, lambda_value=lam)
        out["qg%d" % case], out["qq%d" % case], out["gg%d" % case] = qg, qq, gg
        out["params%d" % case] = np.asarray([k1, k2, lam], dtype=np.float64)
        out["final%d" % case] = final.astype(np.float32)
    out["cases"] = np.asarray(3)
    np.savez_compressed("tests/golden/rerank_golden.npz", **out)
    print("wrote tests/golden/rerank_golden.npz")


if __name__ == "__main__":
    main()
