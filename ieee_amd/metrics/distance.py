"""Distance matrices on MI355X.  Same names, arguments and error behaviour as the
reference's torchreid/metrics/distance.py:6-80; the arithmetic runs in
ieee_sqeuclid_distmat (tiled MFMA GEMM with the norm epilogue fused)."""
import torch

from .. import _lib


def _as_device(x, dtype):
    dev = x.device
    if dev.type != "cuda":
        x = x.cuda(non_blocking=False)
    return x.to(dtype).contiguous(), dev


def _distmat(input1, input2, metric_id, compute_dtype=None):
    lib = _lib.require_gpu()
    if compute_dtype is None:
        compute_dtype = torch.bfloat16 if input1.dtype == torch.bfloat16 else torch.float32
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
    work = torch.empty(m + n, dtype=torch.float32, device=a.device)
    dt = _lib.IEEE_BF16 if compute_dtype == torch.bfloat16 else _lib.IEEE_F32
    _lib.check(lib.ieee_sqeuclid_distmat(_lib.ptr(a), _lib.ptr(b), m, n, d, dt, metric_id, _lib.ptr(out), n,
                                         _lib.ptr(work), _lib.stream()))
    return out if dev.type == "cuda" else out.to(dev)


def compute_distance_matrix(input1, input2, metric='euclidean'):
    """Reference torchreid/metrics/distance.py:6-46.  CPU inputs are staged through the GPU and the
    result is returned on the inputs' device (the reference's caller, Engine._evaluate, passes CPU
    tensors, engine/engine.py:368-399); CUDA inputs stay on the device."""
    assert isinstance(input1, torch.Tensor)
    assert isinstance(input2, torch.Tensor)
    assert input1.dim() == 2, 'Expected 2-D tensor, but got {}-D'.format(input1.dim())
    assert input2.dim() == 2, 'Expected 2-D tensor, but got {}-D'.format(input2.dim())
    assert input1.size(1) == input2.size(1)

    if metric == 'euclidean':
        distmat = euclidean_squared_distance(input1, input2)
    elif metric == 'cosine':
        distmat = cosine_distance(input1, input2)
    else:
        raise ValueError(
            'Unknown distance metric: {}. '
            'Please choose either "euclidean" or "cosine"'.format(metric)
        )
    return distmat


def euclidean_squared_distance(input1, input2):
    """distance.py:49-64 — |a|^2 + |b|^2 - 2 a.b (squared, no sqrt)."""
    return _distmat(input1, input2, 0)


def cosine_distance(input1, input2):
    """distance.py:67-80 — 1 - a^.b^ with F.normalize's eps of 1e-12."""
    return _distmat(input1, input2, 1)
