"""k-reciprocal re-ranking: the oracle restatement is pinned bit-exactly to goldens produced by the
reference's torchreid/utils/rerank.py (CPU test); the device implementation (ieee_rerank) matches the goldens and the
oracle to float rounding and reproduces every discrete decision (GPU tests)."""
import os

import numpy as np
import pytest
import torch

from oracle import rerank as orr

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "rerank_golden.npz"))


def _case(c):
    k1, k2, lam = GOLD["params%d" % c]
    return GOLD["qg%d" % c], GOLD["qq%d" % c], GOLD["gg%d" % c], int(k1), int(k2), float(lam), GOLD["final%d" % c]


def test_oracle_is_bit_exact_against_reference_goldens():
    for c in range(int(GOLD["cases"])):
        qg, qq, gg, k1, k2, lam, want = _case(c)
        got = orr.re_ranking(qg, qq, gg, k1, k2, lam)
        assert got.dtype == np.float32 and np.array_equal(got, want)


def _features(seed, Q, G, D=24, ids=9):
    rng = np.random.RandomState(seed)
    centers = rng.randn(ids, D) * 2.0
    qf = (centers[rng.randint(0, ids, Q)] + rng.randn(Q, D)).astype(np.float32)
    gf = (centers[rng.randint(0, ids, G)] + rng.randn(G, D)).astype(np.float32)
    return qf, gf


@pytest.mark.gpu
def test_device_rerank_matches_goldens():
    from ieee_amd.rerank import re_ranking
    for c in range(int(GOLD["cases"])):
        qg, qq, gg, k1, k2, lam, want = _case(c)
        got = re_ranking(qg, qq, gg, k1=k1, k2=k2, lambda_value=lam)
        assert isinstance(got, np.ndarray) and got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6)
        # the discrete part (which gallery entries have a non-zero Jaccard overlap) is identical
        lam32 = np.float32(lam)
        assert np.array_equal(np.isclose(got, want, rtol=2e-5, atol=2e-6), np.ones_like(want, dtype=bool))
    with pytest.raises(ValueError):
        re_ranking(GOLD["qg0"], GOLD["qq0"][:-1], GOLD["gg0"])


@pytest.mark.gpu
@pytest.mark.parametrize("Q,G,k1,k2", [(150, 700, 20, 6), (64, 64, 8, 1), (333, 1201, 20, 6)])
def test_device_rerank_matches_oracle_on_device_distmats(Q, G, k1, k2):
    """the path Engine._evaluate(rerank=True) takes: distance matrices from the device distmat kernel, re-ranked on the
    device, against the oracle fed with the same matrices; and the re-ranked matrix ranks like the oracle's"""
    from ieee_amd.metrics.distance import compute_distance_matrix
    from ieee_amd.rerank import re_ranking
    qf, gf = _features(Q + G, Q, G)
    q, g = torch.from_numpy(qf).cuda(), torch.from_numpy(gf).cuda()
    qg, qq, gg = compute_distance_matrix(q, g), compute_distance_matrix(q, q), compute_distance_matrix(g, g)
    got = re_ranking(qg, qq, gg, k1=k1, k2=k2, lambda_value=0.3)
    assert got.is_cuda
    want = orr.re_ranking(qg.cpu().numpy(), qq.cpu().numpy(), gg.cpu().numpy(), k1, k2, 0.3)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=5e-5, atol=5e-6)
    # top-10 agreement wherever the oracle's 10th/11th distances are separated by more than the tolerance
    a, b = np.argsort(got.cpu().numpy(), axis=1, kind="stable")[:, :10], np.argsort(want, axis=1, kind="stable")[:, :10]
    sw = np.sort(want, axis=1)
    clear = (np.diff(sw[:, :11], axis=1) > 1e-4).all(axis=1)
    assert clear.sum() > 0 and np.array_equal(a[clear], b[clear])
