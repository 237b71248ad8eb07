"""CPU: the evaluator oracle against the golden vectors captured from the imported reference
(tests/golden/gen_evaluator_golden.py) and against oracle/_ref (the reference's own Cython
evaluator) when that has been built."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import evaluator as ev


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "evaluator_golden.npz"))


def detie(d):
    return (d + (np.arange(d.shape[1], dtype=np.float32) / 1024.0)[None, :]).astype(np.float32)


CASES = [("A", 5, False), ("B", 20, True), ("C", 20, True), ("D", 20, False), ("E", 20, False)]


@pytest.mark.parametrize("tag,max_rank,tie", CASES)
@pytest.mark.parametrize("impl", ["np", "c"])
def test_rank_oracle_matches_reference_golden(G, tag, max_rank, tie, impl):
    d = G[tag + "_dist"]
    if tie:
        d = detie(d)
    f = ev.rank_market1501_np if impl == "np" else ev.rank_market1501_c
    cmc, m_ap = f(d, G[tag + "_qp"], G[tag + "_gp"], G[tag + "_qc"], G[tag + "_gc"], max_rank)
    assert cmc.dtype == np.float32
    assert np.array_equal(cmc, G[tag + "_cmc"])          # bit-exact CMC
    assert abs(m_ap - float(G[tag + "_map"])) < 1e-12


def test_rank_oracle_no_valid_query_raises(G):
    with pytest.raises(AssertionError, match=str(G["G_msg"])):
        ev.rank_market1501_c(G["E_dist"], G["E_qp"] + 100, G["E_gp"], G["E_qc"], G["E_gc"])
    with pytest.raises(AssertionError):
        ev.rank_market1501_np(G["E_dist"], G["E_qp"] + 100, G["E_gp"], G["E_qc"], G["E_gc"])


@pytest.mark.parametrize("tag", ["B", "C"])
def test_distmat_oracle_exact_on_integer_grid(G, tag):
    dm = ev.sqeuclid_np(G[tag + "_qf"].astype(np.float32), G[tag + "_gf"].astype(np.float32))
    assert np.array_equal(dm, G[tag + "_dist"])
    assert np.array_equal(np.argsort(detie(dm)[0], kind="stable").astype(np.int32), G[tag + "_argsort0"])


def test_distmat_oracle_float_tolerance(G):
    q, g = G["F_qf"].astype(np.float32), G["F_gf"].astype(np.float32)
    dm = ev.sqeuclid_np(q, g)
    np.testing.assert_allclose(dm, G["F_dist"], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(ev.cosine_np(q, g), G["F_cos"], rtol=0, atol=1e-6)


def test_oracle_vs_reference_native_evaluator():
    """oracle/_ref/rank_cy*.so is the reference's Cython evaluator built from its own source."""
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
    so = [f for f in (os.listdir(here) if os.path.isdir(here) else []) if f.startswith("rank_cy") and f.endswith(".so")]
    if not so:
        pytest.skip("oracle/_ref not built (needs /root/reference; see oracle/build_ref.sh)")
    spec = importlib.util.spec_from_file_location("rank_cy", os.path.join(here, so[0]))
    rank_cy = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rank_cy)
    rng = np.random.RandomState(7)
    nq, ng = 120, 1500
    d = (rng.rand(nq, ng) * 20).astype(np.float32)
    qp, gp = rng.randint(0, 60, nq), rng.randint(0, 60, ng)
    qc, gc = rng.randint(0, 4, nq), rng.randint(0, 4, ng)
    cmc_ref, map_ref = rank_cy.evaluate_cy(d, qp, gp, qc, gc, 20, False)
    cmc, m_ap = ev.rank_market1501_c(d, qp, gp, qc, gc, 20)
    assert np.array_equal(cmc, cmc_ref)
    assert abs(m_ap - float(map_ref)) < 1e-6      # the Cython path accumulates AP in float32
    cmc32, map32 = ev.rank_market1501_c(d, qp, gp, qc, gc, 20, f32_accum=True)
    assert abs(map32 - float(map_ref)) < 1e-6


def test_split_pieces_sum_exactly_and_kept_products_are_fp32_grade():
    """the split-bf16 restatement: three bf16 pieces reproduce every fp32 value exactly, and the six kept piece
    products differ from the exact product sum by less than fp32 rounding of the result"""
    rng = np.random.RandomState(3)
    x = (rng.randn(64, 768) * np.exp(rng.randn(64, 768) * 3)).astype(np.float32)
    hi, mid, lo = ev.bf16_pieces(x, 3)
    for p in (hi, mid, lo):
        assert np.all((p.view(np.uint32) & 0xFFFF) == 0)           # each piece is a bf16 value
    assert np.array_equal((hi.astype(np.float64) + mid + lo).astype(np.float32), x)
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))
    q = np.abs(rng.randn(50, 768)).astype(np.float32)
    g = np.abs(rng.randn(70, 768)).astype(np.float32)
    q64, g64 = q.astype(np.float64), g.astype(np.float64)
    exact = (q64 ** 2).sum(1)[:, None] + (g64 ** 2).sum(1)[None, :] - 2.0 * (q64 @ g64.T)
    d6 = ev.sqeuclid_split_np(q, g, "bf16x3")
    d3 = ev.sqeuclid_split_np(q, g, "bf16x2")
    dh = ev.sqeuclid_split_np(q, g, "f16x2")
    scale = np.abs(q.astype(np.float64)) @ np.abs(g.astype(np.float64)).T      # sum |q_k g_k|
    assert np.max(np.abs(d6 - exact) / scale) < 2.0 ** -22
    assert np.max(np.abs(d3 - exact) / scale) < 2.0 ** -14
    assert np.max(np.abs(dh - exact) / scale) < 2.0 ** -20
    # rows of very different magnitude: the per-row power-of-two scaling keeps the fp16 pieces in range
    qw = (q * np.exp2(rng.randint(-40, 40, size=(50, 1)))).astype(np.float32)
    qw64 = qw.astype(np.float64)
    exact_w = (qw64 ** 2).sum(1)[:, None] + (g64 ** 2).sum(1)[None, :] - 2.0 * (qw64 @ g64.T)
    scale_w = np.abs(qw64) @ np.abs(g64).T
    assert np.max(np.abs(ev.sqeuclid_split_np(qw, g, "f16x2") - exact_w) / (scale_w + (qw64 ** 2).sum(1)[:, None])) < 2.0 ** -20
    # integer-grid features are single-piece values: nothing is dropped
    qi = rng.randint(-3, 4, size=(9, 24)).astype(np.float32)
    gi = rng.randint(-3, 4, size=(11, 24)).astype(np.float32)
    for scheme in ("bf16x3", "bf16x2", "f16x2"):
        assert np.array_equal(ev.sqeuclid_split_np(qi, gi, scheme), ev.sqeuclid_np(qi, gi).astype(np.float64))
