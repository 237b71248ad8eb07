"""Distance matrices on MI355X.  Same names, arguments and error behaviour as the
reference's torchreid/metrics/distance.py:6-80; the arithmetic runs in
ieee_sqeuclid_distmat (tiled MFMA GEMM with the norm epilogue fused).

`precision` (keyword, or IEEE_DISTMAT_PRECISION; not in the reference) picks the matrix pipe for fp32 inputs:
"fp32" (default) = fp32 MFMA; "bf16x3" = the fp32 rows as three exact bf16 pieces and six piece products on the bf16
matrix cores (fp32-grade accuracy); "f16x2" = two fp16 pieces of the power-of-two-scaled rows and three products
(2^-22 relative, the fastest fp32-grade path); "bf16x2" = two bf16 pieces (~2^-16); "bf16" = round the inputs to bf16."""
import os

import torch

from .. import _lib


def _as_device(x, dtype):
    dev = x.device
    if dev.type != "cuda":
        x = x.cuda(non_blocking=False)
    return x.to(dtype).contiguous(), dev


PRECISIONS = ("fp32", "bf16x3", "f16x2", "bf16x2", "bf16")
_SPLIT_SCHEME = {"bf16x3": 6, "bf16x2": 3, "f16x2": 2}     # IEEE_SPLIT_* in include/ieee_amd.h


def _distmat(input1, input2, metric_id, compute_dtype=None, precision=None):
    lib = _lib.require_gpu()
    if precision is None:
        precision = os.environ.get("IEEE_DISTMAT_PRECISION", "fp32")
    if precision not in PRECISIONS:
        raise ValueError("Unknown distmat precision: {}. Choose one of {}".format(precision, PRECISIONS))
    if compute_dtype is None:
        compute_dtype = torch.bfloat16 if (input1.dtype == torch.bfloat16 or precision == "bf16") else torch.float32
    a, dev = _as_device(input1, compute_dtype)
    b, _ = _as_device(input2, compute_dtype)
    m, d = a.shape
    n = b.shape[0]
    out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    if m == 0 or n == 0:
        return out.to(dev)
    if d % 8 != 0:   # kernels move 16-byte chunks: pad the feature axis with zeros (changes nothing)
        pad = 8 - d % 8
        a = torch.nn.functional.pad(a, (0, pad))
        b = torch.nn.functional.pad(b, (0, pad))
        d += pad
    if compute_dtype == torch.float32 and precision in _SPLIT_SCHEME:
        terms = _SPLIT_SCHEME[precision]
        nbytes = lib.ieee_sqeuclid_distmat_split_workspace_bytes(m, n, d, terms)
        work = torch.empty(nbytes, dtype=torch.uint8, device=a.device)
        _lib.check(lib.ieee_sqeuclid_distmat_split(_lib.ptr(a), _lib.ptr(b), m, n, d, terms, metric_id, _lib.ptr(out), n,
                                                   _lib.ptr(work), nbytes, _lib.stream()))
        return out if dev.type == "cuda" else out.to(dev)
    work = torch.empty(m + n, dtype=torch.float32, device=a.device)
    dt = _lib.IEEE_BF16 if compute_dtype == torch.bfloat16 else _lib.IEEE_F32
    _lib.check(lib.ieee_sqeuclid_distmat(_lib.ptr(a), _lib.ptr(b), m, n, d, dt, metric_id, _lib.ptr(out), n,
                                         _lib.ptr(work), _lib.stream()))
    return out if dev.type == "cuda" else out.to(dev)


def compute_distance_matrix(input1, input2, metric='euclidean', precision=None):
    """Reference torchreid/metrics/distance.py:6-46.  CPU inputs are staged through the GPU and the
    result is returned on the inputs' device (the reference's caller, Engine._evaluate, passes CPU
    tensors, engine/engine.py:368-399); CUDA inputs stay on the device."""
    assert isinstance(input1, torch.Tensor)
    assert isinstance(input2, torch.Tensor)
    assert input1.dim() == 2, 'Expected 2-D tensor, but got {}-D'.format(input1.dim())
    assert input2.dim() == 2, 'Expected 2-D tensor, but got {}-D'.format(input2.dim())
    assert input1.size(1) == input2.size(1)

    if metric == 'euclidean':
        distmat = euclidean_squared_distance(input1, input2, precision)
    elif metric == 'cosine':
        distmat = cosine_distance(input1, input2, precision)
    else:
        raise ValueError(
            'Unknown distance metric: {}. '
            'Please choose either "euclidean" or "cosine"'.format(metric)
        )
    return distmat


def euclidean_squared_distance(input1, input2, precision=None):
    """distance.py:49-64 — |a|^2 + |b|^2 - 2 a.b (squared, no sqrt)."""
    return _distmat(input1, input2, 0, precision=precision)


def cosine_distance(input1, input2, precision=None):
    """distance.py:67-80 — 1 - a^.b^ with F.normalize's eps of 1e-12."""
    return _distmat(input1, input2, 1, precision=precision)
