"""k-reciprocal re-ranking on the device, same call as the reference's
torchreid/utils/rerank.py::re_ranking(q_g_dist, q_q_dist, g_g_dist, k1=20, k2=6, lambda_value=0.3)."""
import numpy as np
import torch

from . import _lib


def _dev(x):
    if isinstance(x, torch.Tensor):
        t = x if x.is_cuda else x.cuda()
    else:
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    return t.to(torch.float32).contiguous()


def re_ranking(q_g_dist, q_q_dist, g_g_dist, k1=20, k2=6, lambda_value=0.3):
    """returns the re-ranked [Q, G] distance matrix: a CUDA tensor when q_g_dist is one, else a numpy array
    (the reference's type, rerank.py:112-113)"""
    lib = _lib.require_gpu()
    qg, qq, gg = _dev(q_g_dist), _dev(q_q_dist), _dev(g_g_dist)
    Q, G = qg.shape
    if qq.shape != (Q, Q) or gg.shape != (G, G):
        raise ValueError("re_ranking: expected q_q_dist %s and g_g_dist %s, got %s and %s"
                         % ((Q, Q), (G, G), tuple(qq.shape), tuple(gg.shape)))
    out = torch.empty((Q, G), dtype=torch.float32, device=qg.device)
    nbytes = int(lib.ieee_rerank_workspace_bytes(Q, G, int(k1)))
    work = torch.empty(nbytes, dtype=torch.uint8, device=qg.device)
    _lib.check(lib.ieee_rerank(_lib.ptr(qg), _lib.ptr(qq), _lib.ptr(gg), Q, G, int(k1), int(k2), float(lambda_value),
                               _lib.ptr(out), _lib.ptr(work), nbytes, _lib.stream()))
    if isinstance(q_g_dist, torch.Tensor) and q_g_dist.is_cuda:
        return out
    return out.cpu().numpy()
