"""Thin tensor-level wrappers over the C ABI (one call = one ieee_* entry point).  Used by the
host-side mirror (model / engine) and by the parity tests; torch only supplies device memory and
the current stream.  Activations are NHWC; a leading group axis (the 3 modalities) is optional."""
import torch

from . import _lib

_DT = {torch.float32: _lib.IEEE_F32, torch.bfloat16: _lib.IEEE_BF16}


def _dt(t):
    return _DT[t.dtype]


def _g(t, nd):
    """returns (groups, group stride in elements) for a tensor with an optional leading group axis"""
    if t.dim() == nd + 1:
        return t.shape[0], t.stride(0)
    return 1, 0


def pack_conv_weight(w, dtype, mode):
    """w: fp32 [Co,Ci,R,S] or [G,Co,Ci,R,S] (reference OIHW) -> packed GEMM operand (see ieee_amd.h)"""
    lib = _lib.require_gpu()
    w = w.contiguous()
    G, gs = _g(w, 4)
    Co, Ci, R, S = w.shape[-4:]
    dt = _DT[dtype]
    ld = lib.ieee_conv_packed_ld(dt, Ci if mode == 0 else Co, R, S)
    rows = Co if mode == 0 else Ci
    shape = (G, rows, ld) if w.dim() == 5 else (rows, ld)
    dst = torch.empty(shape, dtype=dtype, device=w.device)
    _lib.check(lib.ieee_pack_conv_weight(_lib.ptr(w), _lib.ptr(dst), dt, mode, G, Co, Ci, R, S, gs, rows * ld,
                                         _lib.stream()))
    return dst


def conv2d_fwd(x, wp, Co, R, S, stride, pad):
    """x NHWC [N,H,W,Ci] or [G,N,H,W,Ci]; wp packed (mode 0)"""
    lib = _lib.require_gpu()
    x = x.contiguous()
    G, xgs = _g(x, 4)
    N, H, W, Ci = x.shape[-4:]
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
    shape = (G, N, Ho, Wo, Co) if x.dim() == 5 else (N, Ho, Wo, Co)
    y = torch.empty(shape, dtype=x.dtype, device=x.device)
    _lib.check(lib.ieee_conv2d_fwd(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(y), _dt(x), G, N, H, W, Ci, Co, R, S, stride,
                                   pad, xgs, wp.stride(0) if wp.dim() == 3 else 0, N * Ho * Wo * Co, None, _lib.stream()))
    return y


def conv2d_dgrad(dy, wpd, in_hw, Ci, R, S, stride, pad, addend=None):
    lib = _lib.require_gpu()
    dy = dy.contiguous()
    G, gs = _g(dy, 4)
    N, Ho, Wo, Co = dy.shape[-4:]
    H, W = in_hw
    shape = (G, N, H, W, Ci) if dy.dim() == 5 else (N, H, W, Ci)
    dx = torch.empty(shape, dtype=dy.dtype, device=dy.device)
    if addend is not None:
        addend = addend.contiguous()
        assert addend.shape == dx.shape and addend.dtype == dx.dtype
    _lib.check(lib.ieee_conv2d_dgrad(_lib.ptr(dy), _lib.ptr(wpd), _lib.ptr(dx), _lib.ptr(addend), _dt(dy), G, N, H, W,
                                     Ci, Co, R, S, stride, pad, gs, wpd.stride(0) if wpd.dim() == 3 else 0,
                                     N * H * W * Ci, None, None, None, None, 0, 1, _lib.stream()))
    return dx


_FOLD_TICKETS = {}


def _fold_tickets(device):
    """int32 ticket words of ieee_conv2d_wgrad_fold: zero before the first call, left zero by every completed launch"""
    key = (device.type, device.index)
    if key not in _FOLD_TICKETS:
        _FOLD_TICKETS[key] = torch.zeros(_lib.load().ieee_conv2d_wgrad_fold_ticket_words(), dtype=torch.int32, device=device)
    return _FOLD_TICKETS[key]


def conv2d_wgrad(dy, x, R, S, stride, pad, out=None, accumulate=False, fold=False):
    """fold=True: ieee_conv2d_wgrad_fold (split-K slabs folded inside the GEMM launch) instead of GEMM + reduction launch"""
    lib = _lib.require_gpu()
    dy, x = dy.contiguous(), x.contiguous()
    G, dygs = _g(dy, 4)
    N, Ho, Wo, Co = dy.shape[-4:]
    _, H, W, Ci = x.shape[-4:]
    shape = (G, Co, Ci, R, S) if dy.dim() == 5 else (Co, Ci, R, S)
    if out is None:
        out = torch.zeros(shape, dtype=torch.float32, device=dy.device)
    nbytes = lib.ieee_conv2d_wgrad_workspace_bytes(_dt(dy), G, N, Ho, Wo, Ci, Co, R, S)
    work = torch.empty(nbytes, dtype=torch.uint8, device=dy.device)
    if fold:
        _lib.check(lib.ieee_conv2d_wgrad_fold(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(out), _lib.ptr(work),
                                              _lib.ptr(_fold_tickets(dy.device)), _dt(dy), G, N, H, W, Ci, Co, R, S, stride,
                                              pad, dygs, x.stride(0) if x.dim() == 5 else 0, Co * Ci * R * S,
                                              1 if accumulate else 0, _lib.stream()))
        return out
    _lib.check(lib.ieee_conv2d_wgrad(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(out), _lib.ptr(work), _dt(dy), G, N, H, W,
                                     Ci, Co, R, S, stride, pad, dygs, x.stride(0) if x.dim() == 5 else 0,
                                     Co * Ci * R * S, 1 if accumulate else 0, _lib.stream()))
    return out
