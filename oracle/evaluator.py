"""ORACLE (test infrastructure): evaluator restatements.

sqeuclid_np        <- torchreid/metrics/distance.py:49-64 (euclidean_squared_distance)
cosine_np          <- torchreid/metrics/distance.py:67-80
rank_market1501_np <- torchreid/metrics/rank.py:103-171  (eval_market1501, python path)
rank_market1501_c  <- same, compiled C (oracle/rank_oracle.c), for full-size runs
accuracy_np        <- torchreid/metrics/accuracy.py:4-38
bf16_pieces / sqeuclid_split_np: the split-bf16 arithmetic of ieee_sqeuclid_distmat_split (not in the reference; it is
                      a way of computing distance.py:49-64 on the bf16 matrix cores), restated in float64 so that
                      a test separates what the split drops from what fp32 accumulation rounds

Pinned by tests/golden/evaluator_golden.npz, generated from the imported
reference by tests/golden/gen_evaluator_golden.py, and by oracle/_ref (the
reference's own Cython evaluator, built by oracle/build_ref.sh).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _LIB = ctypes.CDLL(path)
        _LIB.ieee_oracle_rank_market1501.restype = ctypes.c_int64
    return _LIB


def sqeuclid_np(q: np.ndarray, g: np.ndarray) -> np.ndarray:
    q = np.asarray(q, np.float32)
    g = np.asarray(g, np.float32)
    mat1 = np.power(q, 2).sum(axis=1, keepdims=True)          # distance.py:60
    mat2 = np.power(g, 2).sum(axis=1, keepdims=True).T        # distance.py:61
    distmat = mat1 + mat2                                     # distance.py:62
    distmat = distmat + np.float32(-2.0) * (q @ g.T)          # distance.py:63 addmm_(beta=1, alpha=-2)
    return distmat.astype(np.float32)


def _bf16_rne(x: np.ndarray) -> np.ndarray:
    """fp32 -> nearest bf16 (ties to even), returned as fp32"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def bf16_pieces(x: np.ndarray, pieces: int = 3):
    """x (fp32) as `pieces` bf16 values hi, mid, lo with hi + mid + lo == x exactly when pieces == 3"""
    x = np.asarray(x, dtype=np.float32)
    out, r = [], x
    for _ in range(pieces):
        p = _bf16_rne(r)
        out.append(p)
        r = (r - p).astype(np.float32)          # exact: the residual fits fp32
    return out


def f16_pieces(x: np.ndarray):
    """rows of x (fp32) scaled by the power of two that puts the row maximum just below 2^14, as two fp16 pieces
    hi, lo (returned as float64, scaled) and the inverse scale per row"""
    x = np.asarray(x, dtype=np.float32)
    mx = np.abs(x).max(axis=1)
    e = np.where(mx > 0, np.frexp(mx)[1], 0).astype(np.int64)
    sc = np.ldexp(np.float32(1.0), (14 - e).astype(np.int32)).astype(np.float32)
    y = (x * sc[:, None]).astype(np.float32)                 # exact (power of two)
    hi = y.astype(np.float16)
    lo = (y - hi.astype(np.float32)).astype(np.float32).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64), (1.0 / sc.astype(np.float64))


def sqeuclid_split_np(q: np.ndarray, g: np.ndarray, scheme="bf16x3") -> np.ndarray:
    """|q|^2 + |g|^2 - 2 q.g with q.g restricted to the piece products the device keeps (float64 sums)"""
    assert scheme in ("bf16x3", "bf16x2", "f16x2")
    if scheme == "f16x2":
        qh, ql, qs = f16_pieces(q)
        gh, gl, gs = f16_pieces(g)
        dot = (ql @ gh.T + qh @ gl.T + qh @ gh.T) * qs[:, None] * gs[None, :]
    else:
        np_ = 3 if scheme == "bf16x3" else 2
        qp, gp = bf16_pieces(q, np_), bf16_pieces(g, np_)
        pairs = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)] if np_ == 3 else [(1, 0), (0, 1), (0, 0)]
        dot = np.zeros((q.shape[0], g.shape[0]), dtype=np.float64)
        for a, b in pairs:
            dot += qp[a].astype(np.float64) @ gp[b].astype(np.float64).T
    qn = (q.astype(np.float64) ** 2).sum(1)[:, None]
    gn = (g.astype(np.float64) ** 2).sum(1)[None, :]
    return qn + gn - 2.0 * dot


def cosine_np(q, g):
    q = np.asarray(q, np.float32)
    g = np.asarray(g, np.float32)
    qn = q / np.maximum(np.sqrt((q * q).sum(1, keepdims=True)), 1e-12)   # F.normalize eps
    gn = g / np.maximum(np.sqrt((g * g).sum(1, keepdims=True)), 1e-12)
    return (1 - qn @ gn.T).astype(np.float32)


def rank_market1501_np(distmat, q_pids, g_pids, q_camids, g_camids, max_rank=20):
    """Pure numpy restatement (small cases).  Ties: (dist, index) ascending."""
    distmat = np.asarray(distmat)
    num_q, num_g = distmat.shape
    if num_g < max_rank:
        max_rank = num_g
    indices = np.argsort(distmat, axis=1, kind="stable")
    matches = (g_pids[indices] == q_pids[:, None]).astype(np.int32)
    all_cmc, all_ap = [], []
    for qi in range(num_q):
        order = indices[qi]
        remove = (g_pids[order] == q_pids[qi]) & (g_camids[order] == q_camids[qi])
        raw = matches[qi][~remove]
        if not np.any(raw):
            continue
        cmc = raw.cumsum()
        cmc[cmc > 1] = 1
        all_cmc.append(cmc[:max_rank])
        num_rel = raw.sum()
        tmp = raw.cumsum() / (np.arange(len(raw)) + 1.0)
        all_ap.append((tmp * raw).sum() / num_rel)
    assert len(all_ap) > 0, "Error: all query identities do not appear in gallery"
    cmc = np.asarray(all_cmc).astype(np.float32).sum(0) / float(len(all_ap))
    return cmc.astype(np.float32), float(np.mean(all_ap))


def rank_market1501_c(distmat, q_pids, g_pids, q_camids, g_camids, max_rank=20, f32_accum=False,
                      return_all_ap=False):
    distmat = np.ascontiguousarray(distmat, np.float32)
    num_q, num_g = distmat.shape
    a = [np.ascontiguousarray(x, np.int64) for x in (q_pids, g_pids, q_camids, g_camids)]
    mr = min(max_rank, num_g)
    cmc = np.zeros(mr, np.float32)
    m_ap = ctypes.c_double(0.0)
    all_ap = np.zeros(num_q, np.float64)
    P = ctypes.c_void_p
    nv = _lib().ieee_oracle_rank_market1501(
        P(distmat.ctypes.data), ctypes.c_int64(num_q), ctypes.c_int64(num_g),
        P(a[0].ctypes.data), P(a[1].ctypes.data), P(a[2].ctypes.data), P(a[3].ctypes.data),
        ctypes.c_int64(max_rank), ctypes.c_int(1 if f32_accum else 0),
        P(cmc.ctypes.data), ctypes.byref(m_ap), P(all_ap.ctypes.data))
    assert nv > 0, "Error: all query identities do not appear in gallery"
    if return_all_ap:
        return cmc, m_ap.value, all_ap
    return cmc, m_ap.value


def accuracy_np(output, target):
    """top-1 accuracy in percent (accuracy.py:25-36 with topk=(1,))."""
    pred = np.argmax(output, axis=1)
    return 100.0 * float((pred == target).sum()) / target.shape[0]
