"""CMC / mAP on MI355X.  Mirrors torchreid/metrics/rank.py:246-287 (evaluate_rank ->
evaluate_py -> eval_market1501 :103-171); the per-query ranking runs in
ieee_rank_market1501 (no sort: the true matches are sorted in LDS and each one's
rank is a prefix sum of a histogram filled while the distance row streams past)."""
import numpy as np
import torch

from .. import _lib


def _i32(x, name, device):
    x = np.asarray(x.cpu() if isinstance(x, torch.Tensor) else x)
    if x.size and (x.max() > np.iinfo(np.int32).max or x.min() < np.iinfo(np.int32).min):
        raise ValueError("%s does not fit int32" % name)
    return torch.from_numpy(np.ascontiguousarray(x.astype(np.int32))).to(device)


def rank_counts(distmat, q_pids, g_pids, q_camids, g_camids, max_rank):
    """the device part of eval_market1501: returns (cmc_counts int64[max_rank], num_valid_q, AP sum) of the given
    queries -- additive over disjoint query sets, which is what a query-sharded evaluation all-reduces"""
    lib = _lib.require_gpu()
    if isinstance(distmat, torch.Tensor):
        d = distmat if distmat.is_cuda else distmat.cuda()
    else:
        d = torch.from_numpy(np.ascontiguousarray(distmat, dtype=np.float32)).cuda()
    d = d.to(torch.float32)
    if d.stride(-1) != 1:
        d = d.contiguous()
    num_q, num_g = d.shape
    dev = d.device
    qp, gp = _i32(q_pids, "q_pids", dev), _i32(g_pids, "g_pids", dev)
    qc, gc = _i32(q_camids, "q_camids", dev), _i32(g_camids, "g_camids", dev)
    assert qp.numel() == num_q and qc.numel() == num_q and gp.numel() == num_g and gc.numel() == num_g
    if num_q == 0:
        return np.zeros(max_rank, dtype=np.int64), 0.0, 0.0
    ap = torch.empty(num_q, dtype=torch.float64, device=dev)
    first = torch.empty(num_q, dtype=torch.int32, device=dev)
    summary = torch.empty(max_rank + 2, dtype=torch.int64, device=dev)
    work = torch.empty(lib.ieee_rank_workspace_bytes(num_g), dtype=torch.uint8, device=dev)
    _lib.check(lib.ieee_rank_market1501_ws(_lib.ptr(d), d.stride(0), num_q, num_g, _lib.ptr(qp), _lib.ptr(gp),
                                           _lib.ptr(qc), _lib.ptr(gc), max_rank, _lib.ptr(ap), _lib.ptr(first),
                                           _lib.ptr(summary), _lib.ptr(work), work.numel(), _lib.stream()))
    s = summary.cpu().numpy()          # the evaluator's single read-back (22 words)
    return s[:max_rank].copy(), float(s[max_rank]), float(s[max_rank + 1:max_rank + 2].view(np.float64)[0])


def finish_counts(cmc_counts, num_valid_q, ap_sum):
    """rank.py:166-169: CMC curve and mAP from the summed per-query results"""
    assert num_valid_q > 0, 'Error: all query identities do not appear in gallery'
    all_cmc = np.asarray(cmc_counts).astype(np.float32) / np.float32(num_valid_q)
    return all_cmc, float(ap_sum / num_valid_q)


def eval_market1501(distmat, q_pids, g_pids, q_camids, g_camids, max_rank):
    """rank.py:103-171.  distmat: numpy array (as the reference's caller passes, engine.py:400) or a
    CUDA tensor (on-device hand-off)."""
    num_g = distmat.shape[1]
    if num_g < max_rank:
        max_rank = num_g
        print('Note: number of gallery samples is quite small, got {}'.format(num_g))
    return finish_counts(*rank_counts(distmat, q_pids, g_pids, q_camids, g_camids, max_rank))


def evaluate_py(distmat, q_pids, g_pids, q_camids, g_camids, max_rank, use_metric_cuhk03):
    if use_metric_cuhk03:
        # the reference's cuhk03 branch is itself broken (rank.py:237-239 passes 6 of 8 arguments)
        raise NotImplementedError("cuhk03 single-gallery-shot protocol is out of scope (SURVEY.md §2 row 8)")
    return eval_market1501(distmat, q_pids, g_pids, q_camids, g_camids, max_rank)


def evaluate_rank(distmat, q_pids, g_pids, q_camids, g_camids, max_rank=20, use_metric_cuhk03=False,
                  use_cython=True):
    """rank.py:246-287.  `use_cython` is accepted and ignored, as in the reference (:278-287)."""
    return evaluate_py(distmat, q_pids, g_pids, q_camids, g_camids, max_rank, use_metric_cuhk03)
