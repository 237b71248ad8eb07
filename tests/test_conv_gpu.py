"""GPU numerics: the implicit-GEMM MFMA convolution (forward / dgrad / wgrad, fp32 and bf16)
against plain torch fp32 conv2d + autograd on the CPU (floating-point kernel: torch fp32 reference)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

#        G  N  H   W   Ci   Co   R  stride pad
CASES = [(1, 2, 16, 8, 64, 256, 1, 1, 0),
         (3, 2, 16, 8, 256, 64, 1, 1, 0),
         (1, 2, 16, 8, 256, 512, 1, 2, 0),
         (3, 1, 16, 8, 64, 64, 3, 1, 1),
         (1, 2, 16, 8, 128, 128, 3, 2, 1),
         (1, 3, 5, 7, 64, 128, 3, 1, 1),       # odd sizes: division (non power-of-two) decode path
         (1, 3, 7, 5, 128, 64, 3, 2, 1),
         (3, 2, 32, 16, 3, 64, 7, 2, 3),       # stem: element-wise gather path
         (1, 1, 8, 8, 512, 2048, 1, 1, 0),
         (1, 4, 16, 8, 128, 128, 1, 1, 0)]


def _ref(x, w, dy, stride, pad):
    x = x.clone().requires_grad_(True)
    w = w.clone().requires_grad_(True)
    y = F.conv2d(x, w, None, stride, pad)
    y.backward(dy)
    return y.detach(), x.grad, w.grad


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_dgrad_wgrad(case, dtype):
    from ieee_amd import _ops
    G, N, H, W, Ci, Co, R, stride, pad = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    rt = (lambda t: t.to(torch.bfloat16).float()) if dtype == torch.bfloat16 else (lambda t: t)
    x = rt(torch.randn(G, N, Ci, H, W, generator=g))
    w = rt(torch.randn(G, Co, Ci, R, R, generator=g) * (2.0 / (Ci * R * R)) ** 0.5)
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    dy = rt(torch.randn(G, N, Co, Ho, Wo, generator=g))
    addend = rt(torch.randn(G, N, Ci, H, W, generator=g))
    refs = [_ref(x[i], w[i], dy[i], stride, pad) for i in range(G)]

    xd = x.permute(0, 1, 3, 4, 2).contiguous().cuda().to(dtype)       # NHWC
    dyd = dy.permute(0, 1, 3, 4, 2).contiguous().cuda().to(dtype)
    wd = w.cuda()
    wp = _ops.pack_conv_weight(wd, dtype, 0)
    y = _ops.conv2d_fwd(xd, wp, Co, R, R, stride, pad).float().cpu().permute(0, 1, 4, 2, 3)
    tol = dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)
    for i in range(G):
        torch.testing.assert_close(y[i], refs[i][0], **tol)

    dw = _ops.conv2d_wgrad(dyd, xd, R, R, stride, pad).cpu()
    wtol = dict(rtol=2e-2, atol=2e-2 * (N * Ho * Wo) ** 0.5) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-3)
    for i in range(G):
        torch.testing.assert_close(dw[i], refs[i][2], **wtol)
    # accumulate flag: dw += once more -> 2x
    dw2 = _ops.conv2d_wgrad(dyd, xd, R, R, stride, pad, out=dw.cuda().clone(), accumulate=True).cpu()
    torch.testing.assert_close(dw2, 2 * dw, rtol=1e-6, atol=1e-6)

    if Ci % 4 == 0:      # the stem needs no dgrad (its input is the image)
        wpd = _ops.pack_conv_weight(wd, dtype, 1)
        dx = _ops.conv2d_dgrad(dyd, wpd, (H, W), Ci, R, R, stride, pad).float().cpu().permute(0, 1, 4, 2, 3)
        dtol = dict(rtol=2e-2, atol=4e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)
        for i in range(G):
            torch.testing.assert_close(dx[i], refs[i][1], **dtol)
        ad = addend.permute(0, 1, 3, 4, 2).contiguous().cuda().to(dtype)
        dx2 = _ops.conv2d_dgrad(dyd, wpd, (H, W), Ci, R, R, stride, pad, addend=ad).float().cpu().permute(0, 1, 4, 2, 3)
        for i in range(G):
            torch.testing.assert_close(dx2[i], refs[i][1] + addend[i], **dtol)


def test_conv_exact_integer_data_fp32():
    """integer-valued data: fp32 MFMA is an exact fma chain, so the result must be bit-identical"""
    from ieee_amd import _ops
    g = torch.Generator().manual_seed(1)
    x = torch.randint(-3, 4, (2, 64, 12, 6), generator=g).float()
    w = torch.randint(-2, 3, (128, 64, 3, 3), generator=g).float()
    y_ref = F.conv2d(x, w, None, 1, 1)
    wp = _ops.pack_conv_weight(w.cuda(), torch.float32, 0)
    y = _ops.conv2d_fwd(x.permute(0, 2, 3, 1).contiguous().cuda(), wp, 128, 3, 3, 1, 1).cpu().permute(0, 3, 1, 2)
    assert torch.equal(y, y_ref)
