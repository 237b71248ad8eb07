"""GPU parity: ieee_sqeuclid_distmat / ieee_rank_market1501 (through the reference-shaped Python
surface, which calls the C ABI) against the reference goldens and the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import evaluator as ev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "evaluator_golden.npz"))


def detie(d):
    return (d + (np.arange(d.shape[1], dtype=np.float32) / 1024.0)[None, :]).astype(np.float32)


@pytest.mark.parametrize("tag", ["B", "C"])
def test_distmat_bit_exact_on_integer_grid(G, tag):
    from ieee_amd.metrics import compute_distance_matrix
    q = torch.from_numpy(G[tag + "_qf"].astype(np.float32))
    g = torch.from_numpy(G[tag + "_gf"].astype(np.float32))
    dm = compute_distance_matrix(q, g, "euclidean")
    assert dm.device.type == "cpu" and dm.dtype == torch.float32       # CPU in -> CPU out, like the reference
    assert np.array_equal(dm.numpy(), G[tag + "_dist"])
    dm2 = compute_distance_matrix(q.cuda(), g.cuda())
    assert dm2.is_cuda and np.array_equal(dm2.cpu().numpy(), G[tag + "_dist"])


def test_distmat_float_and_cosine_tolerance(G):
    from ieee_amd.metrics import compute_distance_matrix
    q = torch.from_numpy(G["F_qf"].astype(np.float32))
    g = torch.from_numpy(G["F_gf"].astype(np.float32))
    dm = compute_distance_matrix(q, g).numpy()
    np.testing.assert_allclose(dm, G["F_dist"], rtol=1e-5, atol=1e-3)   # tolerance: SURVEY §8c, 1e-5 rel
    dc = compute_distance_matrix(q, g, "cosine").numpy()
    np.testing.assert_allclose(dc, G["F_cos"], rtol=0, atol=2e-6)
    with pytest.raises(ValueError):
        compute_distance_matrix(q, g, "manhattan")
    with pytest.raises(AssertionError):
        compute_distance_matrix(q[0], g)


@pytest.mark.parametrize("m,n,d", [(1, 1, 8), (5, 130, 24), (129, 257, 100), (300, 77, 2304), (64, 512, 772)])
def test_distmat_ragged_shapes_vs_oracle(m, n, d):
    from ieee_amd.metrics import compute_distance_matrix
    rng = np.random.RandomState(m * 1000 + n)
    q = rng.randint(-3, 4, size=(m, d)).astype(np.float32)
    g = rng.randint(-3, 4, size=(n, d)).astype(np.float32)
    dm = compute_distance_matrix(torch.from_numpy(q), torch.from_numpy(g)).numpy()
    assert np.array_equal(dm, ev.sqeuclid_np(q, g))      # integer features: exact in any summation order


def test_distmat_bf16_inputs():
    from ieee_amd.metrics import compute_distance_matrix
    rng = np.random.RandomState(3)
    q = torch.from_numpy(np.abs(rng.randn(200, 768)).astype(np.float32)).cuda().bfloat16()
    g = torch.from_numpy(np.abs(rng.randn(333, 768)).astype(np.float32)).cuda().bfloat16()
    dm = compute_distance_matrix(q, g).cpu().numpy()
    ref = ev.sqeuclid_np(q.float().cpu().numpy(), g.float().cpu().numpy())
    np.testing.assert_allclose(dm, ref, rtol=1e-5, atol=2e-3)


@pytest.mark.parametrize("precision", ["bf16x3", "bf16x2", "f16x2"])
def test_distmat_split_bf16_exact_on_integer_grid_and_ragged(G, precision):
    """small integers are single-piece bf16 values: the split GEMM is exact, on the goldens and on ragged shapes"""
    from ieee_amd.metrics import compute_distance_matrix
    for tag in ("B", "C"):
        q = torch.from_numpy(G[tag + "_qf"].astype(np.float32))
        g = torch.from_numpy(G[tag + "_gf"].astype(np.float32))
        assert np.array_equal(compute_distance_matrix(q, g, precision=precision).numpy(), G[tag + "_dist"])
    for m, n, d in [(1, 1, 8), (5, 130, 24), (129, 257, 100), (64, 512, 772)]:
        rng = np.random.RandomState(m * 1000 + n)
        q = rng.randint(-3, 4, size=(m, d)).astype(np.float32)
        g = rng.randint(-3, 4, size=(n, d)).astype(np.float32)
        dm = compute_distance_matrix(torch.from_numpy(q), torch.from_numpy(g), precision=precision).numpy()
        assert np.array_equal(dm, ev.sqeuclid_np(q, g))


@pytest.mark.parametrize("precision", ["bf16x3", "f16x2"])
def test_distmat_split_is_fp32_grade(G, precision):
    """six piece products on the bf16 matrix cores against (a) the reference golden at the fp32 tolerance, (b) the
    float64 restatement of the same piece products -- what is left is fp32 accumulation rounding, and it is no
    larger than the fp32-MFMA path's own --, (c) identical CMC / mAP"""
    from ieee_amd.metrics import compute_distance_matrix, evaluate_rank
    q = torch.from_numpy(G["F_qf"].astype(np.float32))
    g = torch.from_numpy(G["F_gf"].astype(np.float32))
    np.testing.assert_allclose(compute_distance_matrix(q, g, precision=precision).numpy(), G["F_dist"], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(compute_distance_matrix(q, g, "cosine", precision=precision).numpy(), G["F_cos"], rtol=0,
                               atol=2e-6)
    rng = np.random.RandomState(17)
    qf = np.abs(rng.randn(500, 768)).astype(np.float32)
    gf = np.abs(rng.randn(3000, 768)).astype(np.float32)
    q64, g64 = qf.astype(np.float64), gf.astype(np.float64)
    exact = (q64 ** 2).sum(1)[:, None] + (g64 ** 2).sum(1)[None, :] - 2.0 * (q64 @ g64.T)
    d32 = compute_distance_matrix(torch.from_numpy(qf).cuda(), torch.from_numpy(gf).cuda()).cpu().numpy()
    d6 = compute_distance_matrix(torch.from_numpy(qf).cuda(), torch.from_numpy(gf).cuda(), precision=precision).cpu().numpy()
    d3 = compute_distance_matrix(torch.from_numpy(qf).cuda(), torch.from_numpy(gf).cuda(), precision="bf16x2").cpu().numpy()
    e32, e6, e3 = np.abs(d32 - exact).max(), np.abs(d6 - exact).max(), np.abs(d3 - exact).max()
    assert e6 <= 1.5 * e32 + 1e-4, (e6, e32)
    assert np.abs(d6 - ev.sqeuclid_split_np(qf, gf, precision)).max() <= 1.5 * e32 + 1e-4
    assert e3 < 0.05 and np.abs(d3 - ev.sqeuclid_split_np(qf, gf, "bf16x2")).max() <= 1.5 * e32 + 1e-4
    # rows of very different magnitude (the fp16 pieces rely on the per-row scaling)
    qw = (qf[:64] * np.exp2(rng.randint(-30, 30, size=(64, 1)))).astype(np.float32)
    dw = compute_distance_matrix(torch.from_numpy(qw).cuda(), torch.from_numpy(gf).cuda(), precision=precision).cpu().numpy()
    dw32 = compute_distance_matrix(torch.from_numpy(qw).cuda(), torch.from_numpy(gf).cuda()).cpu().numpy()
    np.testing.assert_allclose(dw, dw32, rtol=2e-5, atol=1e-3)
    qp, gp = rng.randint(0, 100, 500), rng.randint(0, 100, 3000)
    qc, gc = rng.randint(0, 4, 500), rng.randint(0, 4, 3000)
    cmc_a, map_a = evaluate_rank(d32, qp, gp, qc, gc)
    cmc_b, map_b = evaluate_rank(d6, qp, gp, qc, gc)
    assert np.array_equal(cmc_a, cmc_b) and abs(map_a - map_b) < 1e-6


def test_distmat_split_errors():
    from ieee_amd import _lib as L
    from ieee_amd.metrics import compute_distance_matrix
    lib = L.load()
    with pytest.raises(ValueError):
        compute_distance_matrix(torch.zeros(2, 8), torch.zeros(2, 8), precision="fp8")
    q = torch.zeros(4, 8, device="cuda")
    out = torch.empty(4, 4, device="cuda")
    work = torch.empty(1 << 16, dtype=torch.uint8, device="cuda")
    assert lib.ieee_sqeuclid_distmat_split_workspace_bytes(4, 4, 8, 5) == -1
    assert lib.ieee_sqeuclid_distmat_split_workspace_bytes(4, 4, 8, 2) > 0
    assert lib.ieee_sqeuclid_distmat_split(L.ptr(q), L.ptr(q), 4, 4, 8, 5, 0, L.ptr(out), 4, L.ptr(work), 1 << 16, L.stream()) != 0
    assert b"scheme" in lib.ieee_last_error()
    assert lib.ieee_sqeuclid_distmat_split(L.ptr(q), L.ptr(q), 4, 4, 8, 6, 0, L.ptr(out), 4, L.ptr(work), 16, L.stream()) != 0
    assert b"workspace" in lib.ieee_last_error()


CASES = [("A", 5, False), ("B", 20, True), ("C", 20, True), ("D", 20, False), ("E", 20, False)]


@pytest.fixture(params=["fast+general", "general-only"])
def rank_path(request, monkeypatch):
    """ieee_rank_market1501 runs rank_query_fast_kernel and leaves the queries that do not fit its LDS lists to
    rank_query_kernel; IEEE_RANK_GENERAL=1 sends every query through the general kernel."""
    if request.param == "general-only":
        monkeypatch.setenv("IEEE_RANK_GENERAL", "1")
    else:
        monkeypatch.delenv("IEEE_RANK_GENERAL", raising=False)
    return request.param


@pytest.mark.parametrize("tag,max_rank,tie", CASES)
def test_rank_matches_reference_golden(G, tag, max_rank, tie, rank_path):
    from ieee_amd.metrics import evaluate_rank
    d = G[tag + "_dist"]
    if tie:
        d = detie(d)
    cmc, m_ap = evaluate_rank(d, G[tag + "_qp"], G[tag + "_gp"], G[tag + "_qc"], G[tag + "_gc"], max_rank=max_rank)
    assert cmc.dtype == np.float32 and isinstance(m_ap, float)
    assert np.array_equal(cmc, G[tag + "_cmc"])          # CMC bit-exact
    assert abs(m_ap - float(G[tag + "_map"])) < 1e-12


def test_rank_no_valid_query_raises(G):
    from ieee_amd.metrics import evaluate_rank
    with pytest.raises(AssertionError, match=str(G["G_msg"])):
        evaluate_rank(G["E_dist"], G["E_qp"] + 100, G["E_gp"], G["E_qc"], G["E_gc"])


@pytest.mark.parametrize("nq,ng,nid,seed", [(300, 3000, 150, 0), (64, 10007, 13, 1), (2000, 20000, 1000, 2)])
def test_rank_vs_oracle_random(nq, ng, nid, seed, rank_path):
    from ieee_amd.metrics import evaluate_rank
    rng = np.random.RandomState(seed)
    d = (rng.rand(nq, ng) * 20).astype(np.float32)
    qp, gp = rng.randint(0, nid, nq), rng.randint(0, nid, ng)
    qc, gc = rng.randint(0, 4, nq), rng.randint(0, 4, ng)
    cmc_o, map_o, ap_o = ev.rank_market1501_c(d, qp, gp, qc, gc, 20, return_all_ap=True)
    cmc, m_ap = evaluate_rank(torch.from_numpy(d).cuda(), qp, gp, qc, gc)
    assert np.array_equal(cmc, cmc_o)
    assert abs(m_ap - map_o) < 1e-12


def test_rank_many_matches_multibatch_and_ties(rank_path):
    """one identity owns 5000 gallery rows (> the 2048-key LDS batch) and distances have exact ties
    (tie order = gallery index, the order the oracle's stable sort defines)."""
    from ieee_amd.metrics import evaluate_rank
    rng = np.random.RandomState(5)
    nq, ng = 16, 9000
    d = rng.randint(0, 50, size=(nq, ng)).astype(np.float32)
    gp = rng.randint(1, 6, ng)
    gp[:5000] = 0
    rng.shuffle(gp)
    qp = np.array([0, 1, 2, 3] * 4)
    qc, gc = rng.randint(0, 3, nq), rng.randint(0, 3, ng)
    cmc_o, map_o = ev.rank_market1501_c(d, qp, gp, qc, gc, 20)
    cmc, m_ap = evaluate_rank(d, qp, gp, qc, gc)
    assert np.array_equal(cmc, cmc_o)
    assert abs(m_ap - map_o) < 1e-12


@pytest.mark.parametrize("case", ["ties", "many_removed", "one_match", "all_equal", "ragged_ld"])
def test_rank_fast_path_edges(case, rank_path):
    """edges of the cell-table kernel: exact distance ties between matches and non-matches (order = gallery index),
    more removed entries than its LDS list holds, a single match (one distance cell), every distance equal, and a
    row stride that is not a multiple of 4 (scalar loads)."""
    from ieee_amd.metrics import evaluate_rank
    rng = np.random.RandomState(21)
    nq, ng = 48, 6151
    d = (rng.rand(nq, ng) * 7).astype(np.float32)
    qp, gp = rng.randint(0, 40, nq), rng.randint(0, 40, ng)
    qc, gc = rng.randint(0, 3, nq), rng.randint(0, 3, ng)
    if case == "ties":
        d = rng.randint(0, 9, size=(nq, ng)).astype(np.float32)
    elif case == "many_removed":
        gp[:3000] = 7
        gc[:2900] = 1
        qp[:8], qc[:8] = 7, 1
        rng.shuffle(gp)
    elif case == "one_match":
        gp[:] = rng.randint(100, 140, ng)
        for i in range(nq):
            gp[i], gc[i], qp[i], qc[i] = 1000 + i, 0, 1000 + i, 1
    elif case == "all_equal":
        d[:] = 3.5
    cmc_o, map_o = ev.rank_market1501_c(d, qp, gp, qc, gc, 20)
    dd = torch.from_numpy(d).cuda()
    if case == "ragged_ld":
        wide = torch.zeros(nq, ng + 3, device="cuda")
        wide[:, :ng] = dd
        dd = wide[:, :ng]
        assert dd.stride(0) % 4 != 0
    cmc, m_ap = evaluate_rank(dd, qp, gp, qc, gc)
    assert np.array_equal(cmc, cmc_o)
    assert abs(m_ap - map_o) < 1e-12


def test_distmat_then_rank_device_pipeline_properties():
    """size-independent properties at a larger size: every query also sits in the gallery under another
    camera at distance 0 -> rank-1 = 1 and AP >= 1/n_match; distmat symmetric & zero diagonal."""
    from ieee_amd.metrics import compute_distance_matrix, evaluate_rank
    rng = np.random.RandomState(11)
    ng, d = 4096, 768
    gf = torch.from_numpy(rng.randint(0, 4, size=(ng, d)).astype(np.float32)).cuda()
    gp = rng.randint(0, 500, ng)
    gc = np.ones(ng, np.int64)
    qf, qp, qc = gf[:512], gp[:512], np.zeros(512, np.int64)
    dm = compute_distance_matrix(qf, gf)
    assert torch.equal(dm[:, :512], dm[:, :512].t())
    assert float(dm[:, :512].diagonal().abs().max()) == 0.0
    cmc, m_ap = evaluate_rank(dm, qp, gp, qc, gc)
    assert cmc[0] == 1.0 and np.all(np.diff(cmc) >= 0) and 0 < m_ap <= 1.0


def test_config4_full_size_sampled_and_property_checks(monkeypatch):
    """BASELINE config 4 at FULL size (10 000 x 100 000 x 768; the 4.0 GB output has byte offsets beyond 2^32), the
    workload bench.py times: 4 096 sampled distmat entries (incl. the corners and the last row) against the oracle's
    float64 evaluation of the reference formula, a planted cross-camera duplicate of every query => rank-1 = 1, a
    monotone CMC, and the same CMC / mAP from the fast and the general ranking kernels."""
    from ieee_amd.metrics import compute_distance_matrix, evaluate_rank
    from oracle import evaluator as ev
    Q, G, D = 10000, 100000, 768
    g = torch.Generator(device="cpu").manual_seed(1)
    qf = torch.randn(Q, D, generator=g).abs()
    gf = torch.randn(G, D, generator=g).abs()
    rs = np.random.RandomState(1)
    qp, gp = rs.randint(0, 1000, Q), rs.randint(0, 1000, G)
    qc, gc = rs.randint(0, 4, Q), rs.randint(0, 4, G)
    plant = rs.permutation(G)[:Q]                      # gallery row plant[i] := query i, same identity, another camera
    gf[plant] = qf
    gp[plant] = qp
    gc[plant] = (qc + 1) % 4
    dm = compute_distance_matrix(qf.cuda(), gf.cuda())
    assert dm.shape == (Q, G) and dm.dtype == torch.float32
    rows = np.concatenate([[0, 0, Q - 1, Q - 1], rs.randint(0, Q, 4092)])
    cols = np.concatenate([[0, G - 1, 0, G - 1], rs.randint(0, G, 4092)])
    got = dm[torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.float64)
    q64, g64 = qf.numpy().astype(np.float64)[rows], gf.numpy().astype(np.float64)[cols]
    ref = (q64 * q64).sum(1) + (g64 * g64).sum(1) - 2.0 * (q64 * g64).sum(1)     # metrics/distance.py:49-64 in float64
    scale = float((q64 * q64).sum(1).max() + (g64 * g64).sum(1).max())
    assert np.abs(got - ref).max() <= 2e-6 * scale                                # fp32 MFMA chain over K = 768
    ref32 = ev.sqeuclid_np(qf.numpy()[rows[:64]], gf.numpy()[cols[:64]])          # the oracle's fp32 sgemm form
    assert np.abs(got[:64] - np.diag(ref32)).max() <= 2e-6 * scale
    planted = dm[torch.arange(Q).cuda(), torch.from_numpy(plant).cuda()].cpu().numpy()
    assert np.abs(planted).max() <= 2e-6 * scale                                  # the duplicate is at distance ~0
    monkeypatch.delenv("IEEE_RANK_GENERAL", raising=False)
    cmc_f, map_f = evaluate_rank(dm, qp, gp, qc, gc)
    monkeypatch.setenv("IEEE_RANK_GENERAL", "1")
    cmc_g, map_g = evaluate_rank(dm, qp, gp, qc, gc)
    assert np.array_equal(cmc_f, cmc_g) and abs(map_f - map_g) < 1e-12
    assert cmc_f[0] == 1.0 and np.all(np.diff(cmc_f) >= 0) and np.all(cmc_f <= 1.0)
    assert 1.0 / 200 < map_f <= 1.0
    del dm
    torch.cuda.empty_cache()


@pytest.mark.parametrize("nq,ng,ids", [(257, 12345, "dense"), (64, 5000, "hashed"), (33, 3000, "collide")])
def test_rank_with_identity_buckets_equals_the_scanning_form(nq, ng, ids):
    """ieee_rank_market1501_ws (gallery bucketed by identity hash once per call; what evaluate_rank uses) against
    ieee_rank_market1501 (every query scans all gallery identities): per-query AP, first-match position and the summary
    must be the same bits.  Identities: small dense ints, arbitrary 32-bit values incl. negatives, and values that
    share a bucket (equal hash, different identity)."""
    from ieee_amd import _lib as L
    lib = L.require_gpu()
    rng = np.random.RandomState(nq)
    if ids == "dense":
        pool = np.arange(200)
    elif ids == "hashed":
        pool = rng.randint(-2**31, 2**31 - 1, size=120, dtype=np.int64)
    else:   # multiples of 2^32 / odd constant do not exist; build collisions by brute force on the kernel's hash
        cand = np.arange(1, 400000, dtype=np.int64)
        h = ((cand * 2654435761) & 0xFFFFFFFF) >> 18
        pool = np.concatenate([cand[h == b][:6] for b in (7, 4097, 16383)])
        assert len(pool) == 18
    gp = rng.choice(pool, ng).astype(np.int32)
    qp = rng.choice(pool, nq).astype(np.int32)
    qc, gc = rng.randint(0, 3, nq).astype(np.int32), rng.randint(0, 3, ng).astype(np.int32)
    d = torch.from_numpy((rng.rand(nq, ng) * 10).astype(np.float32)).cuda()
    dev = [torch.from_numpy(a).cuda() for a in (qp, gp, qc, gc)]
    outs = []
    for ws in (False, True):
        ap = torch.full((nq,), 7.0, dtype=torch.float64, device="cuda")
        first = torch.full((nq,), 7, dtype=torch.int32, device="cuda")
        summ = torch.zeros(22, dtype=torch.int64, device="cuda")
        args = (L.ptr(d), d.stride(0), nq, ng, *[L.ptr(a) for a in dev], 20, L.ptr(ap), L.ptr(first), L.ptr(summ))
        if ws:
            work = torch.empty(lib.ieee_rank_workspace_bytes(ng), dtype=torch.uint8, device="cuda")
            L.check(lib.ieee_rank_market1501_ws(*args, L.ptr(work), work.numel(), L.stream()))
            assert lib.ieee_rank_market1501_ws(*args, L.ptr(work), 16, L.stream()) != 0      # short workspace: refused
        else:
            L.check(lib.ieee_rank_market1501(*args, L.stream()))
        torch.cuda.synchronize()
        outs.append((ap.cpu(), first.cpu(), summ.cpu()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    assert int(outs[0][2][20]) > 0
    cmc_o, map_o = ev.rank_market1501_c(d.cpu().numpy(), qp, gp, qc, gc, 20)
    nv = float(outs[1][2][20])
    assert np.array_equal(outs[1][2][:20].numpy().astype(np.float32) / np.float32(nv), cmc_o)
