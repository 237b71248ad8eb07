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
         (1, 4, 16, 8, 128, 128, 1, 1, 0),
         (2, 2, 32, 16, 128, 128, 3, 2, 1)]    # stride-2 3x3 dgrad in parity-class row order (bf16: dead taps skipped)


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


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("form", ["c8_7x8_pad3", "c4_8x8_prepadded"])
def test_padded_stem_equals_7x7_conv(dtype, form):
    """the stem as a zero-padded vector-path conv: (a) 7x8 over 8 channels with padding 3 (one k-tile = 8 contiguous
    pixels of a row); (b) what the executor uses: 8x8 over 4 channels WITHOUT padding on an image that
    ieee_nchw_to_nhwc3 wrote with a 3-pixel zero border (a 16-byte chunk = 2 adjacent pixels, a k-tile = 2 filter rows)"""
    from ieee_amd import _lib as L
    lib = L.require_gpu()
    g = torch.Generator().manual_seed(7)
    rt = (lambda t: t.to(torch.bfloat16).float()) if dtype == torch.bfloat16 else (lambda t: t)
    G, N, H, W, Co = 3, 2, 32, 16, 64
    x = rt(torch.randn(G, N, 3, H, W, generator=g))
    w = rt(torch.randn(G, Co, 3, 7, 7, generator=g) * 0.1)
    dy = rt(torch.randn(G, N, Co, H // 2, W // 2, generator=g))
    refs = [_ref(x[i], w[i], dy[i], 2, 3) for i in range(G)]
    dt = L.IEEE_BF16 if dtype == torch.bfloat16 else L.IEEE_F32
    xs = [x[i].contiguous().cuda() for i in range(G)]
    if form == "c8_7x8_pad3":
        Cp, Rp, Sp, ipad, cpad = 8, 7, 8, 0, 3
    else:
        Cp, Rp, Sp, ipad, cpad = 4, 8, 8, 3, 0
    Hp, Wp = H + 2 * ipad, W + 2 * ipad
    xp = torch.full((G, N, Hp, Wp, Cp), 7.0, device="cuda", dtype=dtype)
    L.check(lib.ieee_nchw_to_nhwc3(L.ptr(xs[0]), L.ptr(xs[1]), L.ptr(xs[2]), L.ptr(xp), dt, N, 3, H, W, Cp, ipad, L.stream()))
    assert float(xp[..., 3:].abs().max()) == 0.0
    if ipad:
        assert float(xp[:, :, :ipad].abs().max()) == 0.0 and float(xp[:, :, :, -ipad:].abs().max()) == 0.0
        torch.testing.assert_close(xp[:, :, ipad:-ipad, ipad:-ipad, :3].float().cpu(), x.permute(0, 1, 3, 4, 2))
    ld = lib.ieee_conv_packed_ld(dt, Cp, Rp, Sp)
    wp = torch.empty(G, Co, ld, device="cuda", dtype=dtype)
    wd = w.cuda().contiguous()
    L.check(lib.ieee_pack_conv_weight_padded(L.ptr(wd), L.ptr(wp), dt, 0, G, Co, 3, 7, 7, Cp, Rp, Sp, Co * 3 * 49, Co * ld,
                                             L.stream()))
    y = torch.empty(G, N, H // 2, W // 2, Co, device="cuda", dtype=dtype)
    L.check(lib.ieee_conv2d_fwd(L.ptr(xp), L.ptr(wp), L.ptr(y), dt, G, N, Hp, Wp, Cp, Co, Rp, Sp, 2, cpad, xp[0].numel(),
                                Co * ld, y[0].numel(), None, L.stream()))
    tol = dict(rtol=2e-2, atol=3e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)
    for i in range(G):
        torch.testing.assert_close(y[i].float().cpu().permute(0, 3, 1, 2), refs[i][0], **tol)
    dyd = dy.permute(0, 1, 3, 4, 2).contiguous().cuda().to(dtype)
    nbytes = lib.ieee_conv2d_wgrad_workspace_bytes(dt, G, N, H // 2, W // 2, Cp, Co, Rp, Sp)
    work = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    dwp = torch.zeros(G, Co, Cp, Rp, Sp, device="cuda")
    L.check(lib.ieee_conv2d_wgrad(L.ptr(dyd), L.ptr(xp), L.ptr(dwp), L.ptr(work), dt, G, N, Hp, Wp, Cp, Co, Rp, Sp, 2, cpad,
                                  dyd[0].numel(), xp[0].numel(), dwp[0].numel(), 0, L.stream()))
    dw = torch.zeros(G, Co, 3, 7, 7, device="cuda")
    L.check(lib.ieee_unpad_weight_grad(L.ptr(dwp), L.ptr(dw), G, Co, Cp, Rp, Sp, 3, 7, 7, dwp[0].numel(), dw[0].numel(), 0,
                                       L.stream()))
    wtol = dict(rtol=2e-2, atol=0.5) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-3)
    for i in range(G):
        torch.testing.assert_close(dw[i].cpu(), refs[i][2], **wtol)


@pytest.mark.parametrize("N,H", [(2, 32), (3, 16), (3, 24), (1, 4)])
def test_direct_stem_persistent_over_four_tiles(N, H):
    """the direct stem kernel at the executor's width (W = 128 -> 64 output columns): bf16, 8x8 / stride 2 over the
    border-padded 4-channel image, with the fused BatchNorm sums.  H = 32 / 16: the workgroup walks 4 tiles (8 output rows)
    with the weights in registers and the next patch prefetched; H = 24 / 4 (6 tiles / 1 tile per image: not a multiple of 4): the
    one-tile-per-workgroup form.  Against torch's fp32 conv on the bf16-rounded operands, and the per-tile sums against
    the stored output."""
    from ieee_amd import _lib as L
    lib = L.require_gpu()
    g = torch.Generator().manual_seed(17)
    rt = lambda t: t.to(torch.bfloat16).float()
    G, W, Co = 3, 128, 64
    x = rt(torch.randn(G, N, 3, H, W, generator=g))
    w = rt(torch.randn(G, Co, 3, 7, 7, generator=g) * 0.1)
    xs = [x[i].contiguous().cuda() for i in range(G)]
    Hp, Wp = H + 6, W + 6
    xp = torch.empty((G, N, Hp, Wp, 4), device="cuda", dtype=torch.bfloat16)
    L.check(lib.ieee_nchw_to_nhwc3(L.ptr(xs[0]), L.ptr(xs[1]), L.ptr(xs[2]), L.ptr(xp), L.IEEE_BF16, N, 3, H, W, 4, 3, L.stream()))
    ld = lib.ieee_conv_packed_ld(L.IEEE_BF16, 4, 8, 8)
    wp = torch.empty(G, Co, ld, device="cuda", dtype=torch.bfloat16)
    wd = w.cuda().contiguous()
    L.check(lib.ieee_pack_conv_weight_padded(L.ptr(wd), L.ptr(wp), L.IEEE_BF16, 0, G, Co, 3, 7, 7, 4, 8, 8, Co * 3 * 49, Co * ld, L.stream()))
    Ho, Wo = H // 2, W // 2
    rb = lib.ieee_conv2d_fwd_stats_rblocks(N, Ho, Wo)
    for stats in (False, True):
        y = torch.full((G, N, Ho, Wo, Co), 9.0, device="cuda", dtype=torch.bfloat16)
        part = torch.zeros(G, 2, Co, rb, device="cuda")
        L.check(lib.ieee_conv2d_fwd(L.ptr(xp), L.ptr(wp), L.ptr(y), L.IEEE_BF16, G, N, Hp, Wp, 4, Co, 8, 8, 2, 0, xp[0].numel(), Co * ld,
                                    y[0].numel(), L.ptr(part) if stats else None, L.stream()))
        for i in range(G):
            ref = torch.nn.functional.conv2d(x[i], w[i], None, 2, 3).permute(0, 2, 3, 1)
            torch.testing.assert_close(y[i].float().cpu(), ref, rtol=2e-2, atol=3e-2)
        if stats:
            yf = y.float().view(G, -1, Co)
            torch.testing.assert_close(part[:, 0].sum(-1), yf.sum(1), rtol=1e-4, atol=2e-2)
            torch.testing.assert_close(part[:, 1].sum(-1), (yf * yf).sum(1), rtol=1e-4, atol=2e-2)
            # per tile (128 rows = 2 output rows), not only in total: a walking workgroup must file each tile's sums under its own index
            tiles = yf.view(G, rb, 128, Co)
            torch.testing.assert_close(part[:, 0].permute(0, 2, 1), tiles.sum(2), rtol=1e-4, atol=2e-2)


def test_fused_bn_partials_in_conv_epilogues():
    """bf16: the forward epilogue's per-tile BN sums (sum y, sum y^2) and the dgrad epilogue's BN-backward sums
    (sum g, sum g*y with the three mask sources) against direct sums over the tensors the kernels stored"""
    from ieee_amd import _lib as L, _ops
    lib = L.require_gpu()
    g = torch.Generator().manual_seed(21)
    G, N, H, W, Ci, Co = 3, 5, 12, 10, 64, 192          # M = 600 rows: 5 row tiles, the last one partial
    dt = torch.bfloat16
    x = torch.randn(G, N, H, W, Ci, generator=g).cuda().to(dt)
    w = (torch.randn(G, Co, Ci, 3, 3, generator=g) * 0.05).cuda()
    wp, wpd = _ops.pack_conv_weight(w, dt, 0), _ops.pack_conv_weight(w, dt, 1)
    M = N * H * W
    rb = lib.ieee_conv2d_fwd_stats_rblocks(N, H, W)
    part = torch.zeros(G, 2, Co, rb, device="cuda")
    y = torch.empty(G, N, H, W, Co, device="cuda", dtype=dt)
    L.check(lib.ieee_conv2d_fwd(L.ptr(x), L.ptr(wp), L.ptr(y), L.IEEE_BF16, G, N, H, W, Ci, Co, 3, 3, 1, 1, x[0].numel(),
                                wp.stride(0), y[0].numel(), L.ptr(part), L.stream()))
    yf = y.float().view(G, M, Co)
    torch.testing.assert_close(part[:, 0].sum(-1), yf.sum(1), rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(part[:, 1].sum(-1), (yf * yf).sum(1), rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(y.float(), _ops.conv2d_fwd(x, wp, Co, 3, 3, 1, 1).float())   # same stores as unfused
    # dgrad: dx = grad w.r.t. x; pretend x = relu(bn(ypre)) of a previous unit
    dy = torch.randn(G, N, H, W, Co, generator=g).cuda().to(dt)
    ypre = torch.randn(G, N, H, W, Ci, generator=g).cuda().to(dt)
    amask = (torch.randn(G, N, H, W, Ci, generator=g) > 0).cuda().to(dt)
    stats = torch.randn(G, 4, Ci, generator=g).cuda()
    addend = torch.randn(G, N, H, W, Ci, generator=g).cuda().to(dt)
    dx_ref = _ops.conv2d_dgrad(dy, wpd, (H, W), Ci, 3, 3, 1, 1, addend=addend)
    # the same mask as packed bits (bit e of byte k = element 8k + e), the form ieee_bn2d_fwd(relu_bits) writes
    mbits = ((amask > 0).view(-1, 8).to(torch.int32) << torch.arange(8, device="cuda", dtype=torch.int32)).sum(1).to(torch.uint8)
    results = {}
    for mode in ("mask", "bits", "stats", "none"):
        p2 = torch.zeros(G, 2, Ci, rb, device="cuda")
        dx = torch.empty_like(dx_ref)
        mptr = L.ptr(amask) if mode == "mask" else (L.ptr(mbits) if mode == "bits" else None)
        L.check(lib.ieee_conv2d_dgrad(L.ptr(dy), L.ptr(wpd), L.ptr(dx), L.ptr(addend), L.IEEE_BF16, G, N, H, W, Ci, Co, 3, 3,
                                      1, 1, dy[0].numel(), wpd.stride(0), dx[0].numel(), L.ptr(p2), L.ptr(ypre),
                                      mptr, L.ptr(stats) if mode == "stats" else None, int(mode == "bits"), 1, L.stream()))
        results[mode] = (dx, p2)
        if mode == "bits":   # bit-identical to the mask-tensor form, stores and sums
            assert torch.equal(dx, results["mask"][0]) and torch.equal(p2, results["mask"][1])
            continue
        # mask-tensor mode stores the MASKED gradient g = (dgrad + addend) * [mask > 0] (what the BatchNorm backward of
        # the block output consumes); the other two modes store dgrad + addend unchanged
        if mode == "mask":
            assert torch.equal(dx, dx_ref * (amask > 0).to(dt))
        else:
            assert torch.equal(dx, dx_ref)
        d, yv = dx_ref.float().view(G, M, Ci), ypre.float().view(G, M, Ci)
        if mode == "mask":
            gq = d * (amask.float().view(G, M, Ci) > 0)
        elif mode == "stats":
            gq = d * ((yv * stats[:, 2:3] + stats[:, 3:4]) > 0)
        else:
            gq = d
        torch.testing.assert_close(p2[:, 0].sum(-1), gq.sum(1), rtol=1e-3, atol=5e-2)
        torch.testing.assert_close(p2[:, 1].sum(-1), (gq * yv).sum(1), rtol=1e-3, atol=5e-2)


@pytest.mark.parametrize("Ci,Co,R", [(256, 64, 1), (64, 192, 3), (1024, 256, 1)])
def test_dgrad_epilogue_sums_for_a_second_batchnorm(Ci, Co, R):
    """ieee_conv2d_dgrad2: the gradient a block-input dgrad produces feeds TWO BatchNorms of the previous block (bn3 and the
    downsample branch's); the epilogue emits sum g / sum g*y for the first as before and sum g / sum g*y2 for the second into
    its own partial block.  Stores and the first block's sums must be the bits of ieee_conv2d_dgrad; the second block's sums
    are checked against direct sums (and plane 0 against the first block's plane 0, bit for bit)."""
    from ieee_amd import _lib as L, _ops
    lib = L.require_gpu()
    g = torch.Generator().manual_seed(33 + Ci)
    G, N, H, W = 3, 5, 12, 10            # M = 600 rows: 5 row tiles, the last one partial
    dt = torch.bfloat16
    pad = R // 2
    w = (torch.randn(G, Co, Ci, R, R, generator=g) * 0.05).cuda()
    wpd = _ops.pack_conv_weight(w, dt, 1)
    M = N * H * W
    rb = lib.ieee_conv2d_fwd_stats_rblocks(N, H, W)
    dy = torch.randn(G, N, H, W, Co, generator=g).cuda().to(dt)
    y1 = torch.randn(G, N, H, W, Ci, generator=g).cuda().to(dt)
    y2 = torch.randn(G, N, H, W, Ci, generator=g).cuda().to(dt)
    amask = (torch.randn(G, N, H, W, Ci, generator=g) > 0).cuda().to(dt)
    mbits = ((amask > 0).view(-1, 8).to(torch.int32) << torch.arange(8, device="cuda", dtype=torch.int32)).sum(1).to(torch.uint8)
    addend = torch.randn(G, N, H, W, Ci, generator=g).cuda().to(dt)
    args = (L.ptr(dy), L.ptr(wpd), None, L.ptr(addend), L.IEEE_BF16, G, N, H, W, Ci, Co, R, R, 1, pad, dy[0].numel(),
            wpd.stride(0), addend[0].numel())
    out = {}
    for two in (False, True):
        dx = torch.empty_like(addend)
        p1 = torch.full((G, 2, Ci, rb), 7.0, device="cuda")
        p2 = torch.full((G, 2, Ci, rb), 7.0, device="cuda")
        a = list(args); a[2] = L.ptr(dx)
        tail = (L.ptr(p1), L.ptr(y1), L.ptr(mbits), None, 1, 1)
        if two:
            L.check(lib.ieee_conv2d_dgrad2(*a, *tail, L.ptr(y2), L.ptr(p2), L.stream()))
        else:
            L.check(lib.ieee_conv2d_dgrad(*a, *tail, L.stream()))
        torch.cuda.synchronize()
        out[two] = (dx, p1, p2)
    assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1])
    assert float(out[False][2].min()) == 7.0                       # the plain call never touches a second block
    p2 = out[True][2]
    assert torch.equal(p2[:, 0], out[True][1][:, 0])
    gq = out[True][0].float().view(G, M, Ci)                        # the stored masked gradient
    torch.testing.assert_close(p2[:, 1].sum(-1), (gq * y2.float().view(G, M, Ci)).sum(1), rtol=1e-3, atol=5e-2)
    # one without the other is refused
    dx = torch.empty_like(addend)
    a = list(args); a[2] = L.ptr(dx)
    assert lib.ieee_conv2d_dgrad2(*a, L.ptr(out[True][1]), L.ptr(y1), L.ptr(mbits), None, 1, 1, L.ptr(y2), None, L.stream()) != 0


def test_dgrad_with_compact_stride2_addend():
    """the block-input dgrad whose identity-branch gradient comes from a stride-2 1x1 downsample conv: the compact
    [N, H/2, W/2, C] addend (addend_stride = 2) gives bit for bit the stores and BatchNorm sums of the full-size map that
    is zero at every pixel with an odd row or column"""
    from ieee_amd import _lib as L, _ops
    lib = L.require_gpu()
    g = torch.Generator().manual_seed(4)
    G, N, H, W, Ci, Co = 3, 3, 16, 8, 256, 64
    dt = torch.bfloat16
    dy = torch.randn(G, N, H, W, Co, generator=g).cuda().to(dt)
    w = (torch.randn(G, Co, Ci, 1, 1, generator=g) * 0.05).cuda()
    wpd = _ops.pack_conv_weight(w, dt, 1)
    ypre = torch.randn(G, N, H, W, Ci, generator=g).cuda().to(dt)
    mask = torch.rand(G, N, H, W, Ci, generator=g).cuda() > 0.5
    bits = (mask.view(-1, 8).to(torch.int32) << torch.arange(8, device="cuda", dtype=torch.int32)).sum(1).to(torch.uint8)
    compact = torch.randn(G, N, H // 2, W // 2, Ci, generator=g).cuda().to(dt)
    full = torch.zeros(G, N, H, W, Ci, device="cuda", dtype=dt)
    full[:, :, ::2, ::2] = compact
    rb = lib.ieee_conv2d_fwd_stats_rblocks(N, H, W)
    got = []
    for addend, stride in ((full, 1), (compact, 2)):
        p2 = torch.zeros(G, 2, Ci, rb, device="cuda")
        dx = torch.empty(G, N, H, W, Ci, device="cuda", dtype=dt)
        L.check(lib.ieee_conv2d_dgrad(L.ptr(dy), L.ptr(wpd), L.ptr(dx), L.ptr(addend), L.IEEE_BF16, G, N, H, W, Ci, Co, 1, 1,
                                      1, 0, dy[0].numel(), wpd.stride(0), dx[0].numel(), L.ptr(p2), L.ptr(ypre), L.ptr(bits),
                                      None, 1, stride, L.stream()))
        got.append((dx, p2))
    assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1])
    plain = _ops.conv2d_dgrad(dy, wpd, (H, W), Ci, 1, 1, 1, 0, addend=full)
    assert torch.equal(got[1][0], plain * mask.to(dt))
    with pytest.raises(L.IeeeAmdError):   # the compact form exists for the fused bf16 epilogue only
        lib_call = lib.ieee_conv2d_dgrad(L.ptr(dy), L.ptr(wpd), L.ptr(dx), L.ptr(compact), L.IEEE_BF16, G, N, H, W, Ci, Co, 1,
                                         1, 1, 0, dy[0].numel(), wpd.stride(0), dx[0].numel(), None, None, None, None, 0, 2,
                                         L.stream())
        L.check(lib_call)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("res,relu", [(False, 1), (True, 1), (False, 0)])
def test_conv_with_fused_eval_batchnorm(dtype, res, relu):
    """inference: conv + BatchNorm(running statistics) (+ residual) (+ ReLU) in one launch (ieee_conv2d_fwd_bn_eval)
    against torch's conv2d -> batch_norm(eval) -> add -> relu"""
    from ieee_amd import _lib as L, _ops
    lib = L.require_gpu()
    g = torch.Generator().manual_seed(13)
    G, N, H, W, Ci, Co = 3, 3, 10, 6, 64, 192
    rt = (lambda t: t.to(torch.bfloat16).float()) if dtype == torch.bfloat16 else (lambda t: t)
    x = rt(torch.randn(G, N, Ci, H, W, generator=g))
    w = rt(torch.randn(G, Co, Ci, 3, 3, generator=g) * 0.05)
    idn = rt(torch.randn(G, N, Co, H, W, generator=g))
    gamma, beta = torch.rand(G, Co, generator=g) + 0.5, torch.randn(G, Co, generator=g) * 0.1
    rm, rv = torch.randn(G, Co, generator=g) * 0.1, torch.rand(G, Co, generator=g) + 0.5
    want = []
    for i in range(G):
        y = F.batch_norm(F.conv2d(x[i], w[i], padding=1), rm[i], rv[i], gamma[i], beta[i], training=False, eps=1e-5)
        if res:
            y = y + idn[i]
        want.append(torch.relu(y) if relu else y)
    dt = L.IEEE_BF16 if dtype == torch.bfloat16 else L.IEEE_F32
    xd = x.permute(0, 1, 3, 4, 2).contiguous().cuda().to(dtype)
    rd = idn.permute(0, 1, 3, 4, 2).contiguous().cuda().to(dtype)
    wp = _ops.pack_conv_weight(w.cuda(), dtype, 0)
    stats = torch.empty(G, 4, Co, device="cuda")
    dummy = torch.empty(G, N, H, W, Co, device="cuda", dtype=dtype)
    part = torch.empty(64, device="cuda")
    gd, bd, rmd, rvd = gamma.cuda(), beta.cuda(), rm.cuda(), rv.cuda()
    L.check(lib.ieee_bn2d_fwd(L.ptr(dummy), None, None, dt, G, N * H * W, Co, N * H * W * Co, L.ptr(gd), L.ptr(bd), Co, L.ptr(rmd),
                              L.ptr(rvd), Co, L.ptr(stats), L.ptr(part), 0.1, 1e-5, 0, relu, 0, None, L.stream()))
    out = torch.empty_like(dummy)
    L.check(lib.ieee_conv2d_fwd_bn_eval(L.ptr(xd), L.ptr(wp), L.ptr(out), L.ptr(rd) if res else None, L.ptr(stats), relu, dt, G, N,
                                        H, W, Ci, Co, 3, 3, 1, 1, xd[0].numel(), wp.stride(0), out[0].numel(), L.stream()))
    tol = dict(rtol=2e-2, atol=3e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)
    for i in range(G):
        torch.testing.assert_close(out[i].float().cpu().permute(0, 3, 1, 2), want[i], **tol)


@pytest.mark.parametrize("R,Ci,Co,H,W", [(1, 128, 320, 16, 8), (3, 64, 128, 16, 8), (3, 128, 64, 8, 16)])
def test_conv_with_fused_train_batchnorm_finalize(R, Ci, Co, H, W):
    """ieee_conv2d_fwd_bn_train (the BatchNorm finalize done by the last-arriving workgroup of each column block) against
    the two-launch form -- ieee_conv2d_fwd with partial sums + ieee_bn2d_fwd's finalize: same y bit for bit, the same
    statistics / scale / shift / running statistics up to the summation order of the per-tile partials (float64 on both
    sides), tickets left at zero; twice in a row (the second launch reuses the tickets the first one reset)."""
    from ieee_amd import _lib as L, _ops
    lib = L.require_gpu()
    g = torch.Generator().manual_seed(R * 100 + Co)
    G, N = 3, 20                                        # M = 2560 rows: 20 row tiles
    dt = torch.bfloat16
    pad = R // 2
    x = torch.randn(G, N, H, W, Ci, generator=g).cuda().to(dt)
    w = (torch.randn(G, Co, Ci, R, R, generator=g) * (2.0 / (Ci * R * R)) ** 0.5).cuda()
    wp = _ops.pack_conv_weight(w, dt, 0)
    M = N * H * W
    assert M <= lib.ieee_conv2d_fwd_bn_train_max_rows()
    rb = lib.ieee_conv2d_fwd_stats_rblocks(N, H, W)
    gam, bet = torch.rand(G, Co, generator=g).cuda() + 0.5, torch.randn(G, Co, generator=g).cuda()
    rm0, rv0 = torch.randn(G, Co, generator=g).cuda(), torch.rand(G, Co, generator=g).cuda() + 0.5
    # two launches
    part = torch.zeros(G, 2, Co, rb, device="cuda")
    y_a = torch.empty(G, N, H, W, Co, device="cuda", dtype=dt)
    L.check(lib.ieee_conv2d_fwd(L.ptr(x), L.ptr(wp), L.ptr(y_a), L.IEEE_BF16, G, N, H, W, Ci, Co, R, R, 1, pad, x[0].numel(),
                                wp.stride(0), y_a[0].numel(), L.ptr(part), L.stream()))
    st_a, rm_a, rv_a = torch.zeros(G, 4, Co, device="cuda"), rm0.clone(), rv0.clone()
    L.check(lib.ieee_bn2d_fwd(L.ptr(y_a), None, None, L.IEEE_BF16, G, M, Co, M * Co, L.ptr(gam), L.ptr(bet), Co, L.ptr(rm_a),
                              L.ptr(rv_a), Co, L.ptr(st_a), L.ptr(part), 0.1, 1e-5, 1, 1, rb, None, L.stream()))
    # one launch, twice
    tickets = torch.zeros(G * ((Co + 63) // 64), dtype=torch.int32, device="cuda")
    for rep in range(2):
        part_b = torch.zeros_like(part)
        y_b = torch.empty_like(y_a)
        st_b, rm_b, rv_b = torch.zeros_like(st_a), rm0.clone(), rv0.clone()
        L.check(lib.ieee_conv2d_fwd_bn_train(L.ptr(x), L.ptr(wp), L.ptr(y_b), L.IEEE_BF16, G, N, H, W, Ci, Co, R, R, 1, pad,
                                             x[0].numel(), wp.stride(0), y_b[0].numel(), L.ptr(part_b), L.ptr(gam), L.ptr(bet), Co,
                                             L.ptr(rm_b), L.ptr(rv_b), Co, L.ptr(st_b), 0.1, 1e-5, L.ptr(tickets), L.stream()))
        torch.cuda.synchronize()
        assert torch.equal(y_a, y_b) and torch.equal(part, part_b)
        assert int(tickets.abs().sum()) == 0
        torch.testing.assert_close(st_b, st_a, rtol=2e-6, atol=1e-6)
        torch.testing.assert_close(rm_b, rm_a, rtol=2e-6, atol=1e-7)
        torch.testing.assert_close(rv_b, rv_a, rtol=2e-6, atol=1e-7)
    # and the statistics are the batch statistics of the stored y
    yf = y_a.float().view(G, M, Co)
    torch.testing.assert_close(st_a[:, 0], yf.mean(1), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_deferred_wgrad_with_one_batched_reduction_is_bit_identical(dtype):
    """ieee_conv2d_wgrad_deferred + ONE ieee_wgrad_reduce_batch over several layers (what the executor does per backward
    part) must give the bits of the immediate ieee_conv2d_wgrad: same slabs, same fixed summation order.  Covers the four
    reduction forms: in place (no slab), 16-byte split-lane (1x1), LDS-transposed (3x3), scalar (odd shapes / stem)."""
    import ctypes
    from ieee_amd import _lib as L
    lib = L.require_gpu()
    g = torch.Generator().manual_seed(11)
    dt = L.IEEE_BF16 if dtype == torch.bfloat16 else L.IEEE_F32
    #        N   H   W   Ci   Co   R  stride pad
    layers = [(8, 16, 8, 256, 64, 1, 1, 0), (8, 16, 8, 64, 64, 3, 1, 1), (8, 16, 8, 512, 512, 3, 1, 1), (16, 8, 8, 1024, 256, 1, 1, 0),
              (2, 16, 8, 256, 512, 1, 2, 0), (3, 5, 7, 64, 128, 3, 1, 1), (8, 32, 16, 8, 64, 7, 2, 3), (64, 32, 16, 64, 256, 1, 1, 0)]
    G = 3
    descs, keep, want = [], [], []
    for (N, H, W, Ci, Co, R, stride, pad) in layers:
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
        x = torch.randn(G, N, H, W, Ci, generator=g).cuda().to(dtype)
        dy = torch.randn(G, N, Ho, Wo, Co, generator=g).cuda().to(dtype)
        nbytes = lib.ieee_conv2d_wgrad_workspace_bytes(dt, G, N, Ho, Wo, Ci, Co, R, R)
        args = (dt, G, N, H, W, Ci, Co, R, R, stride, pad, dy[0].numel(), x[0].numel(), Co * Ci * R * R)
        for acc in (0, 1):
            base = torch.randn(G, Co, Ci, R, R, generator=g).cuda()
            ref, out = base.clone(), base.clone()
            work0 = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            L.check(lib.ieee_conv2d_wgrad(L.ptr(dy), L.ptr(x), L.ptr(ref), L.ptr(work0), *args, acc, L.stream()))
            work = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            d = L.WgradReduceDesc()
            L.check(lib.ieee_conv2d_wgrad_deferred(L.ptr(dy), L.ptr(x), L.ptr(out), L.ptr(work), *args, acc, ctypes.addressof(d),
                                                   L.stream()))
            keep.append((x, dy, work, out)); want.append(ref)
            if d.kind != 0:
                descs.append(d)
    kinds = {d.kind for d in descs}
    assert {1, 3} <= kinds <= {1, 2, 3}, kinds
    assert len(descs) < len(want)          # at least one gradient was written in place (kind 0)
    tab = (L.WgradReduceDesc * len(descs))()
    blocks = 0
    for i, d in enumerate(descs):
        ctypes.memmove(ctypes.addressof(tab[i]), ctypes.addressof(d), ctypes.sizeof(d))
        tab[i].block_begin = blocks
        blocks += d.blocks
    raw = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).cuda()
    L.check(lib.ieee_wgrad_reduce_batch(L.ptr(raw), len(descs), blocks, G, L.stream()))
    torch.cuda.synchronize()
    for (x, dy, work, out), ref in zip(keep, want):
        assert torch.equal(out, ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_chained_wgrad_runs_the_previous_reduction_as_its_prologue_bit_identical(dtype):
    """ieee_conv2d_wgrad_chained: every launch folds the slabs of the PREVIOUS layer before its own GEMM (two alternating slab
    regions), the last descriptor is flushed by ieee_wgrad_reduce_pending.  Same kernels' arithmetic as the immediate
    ieee_conv2d_wgrad -> the same bits, for every reduction form, with and without `accumulate`, and a refused call when the
    pending reduction still reads the region handed in as `work`."""
    import ctypes
    from ieee_amd import _lib as L
    lib = L.require_gpu()
    g = torch.Generator().manual_seed(12)
    dt = L.IEEE_BF16 if dtype == torch.bfloat16 else L.IEEE_F32
    #        N   H   W   Ci   Co   R  stride pad
    layers = [(8, 16, 8, 256, 64, 1, 1, 0), (8, 16, 8, 64, 64, 3, 1, 1), (8, 16, 8, 512, 512, 3, 1, 1), (16, 8, 8, 1024, 256, 1, 1, 0),
              (2, 16, 8, 256, 512, 1, 2, 0), (3, 5, 7, 64, 128, 3, 1, 1), (2, 70, 134, 4, 64, 8, 2, 0), (8, 32, 16, 8, 64, 7, 2, 3),
              (64, 32, 16, 64, 256, 1, 1, 0), (64, 16, 8, 2048, 2048, 1, 1, 0), (64, 16, 8, 256, 256, 3, 1, 1)]
    G = 3
    big = max(lib.ieee_conv2d_wgrad_workspace_bytes(dt, G, N, (H + 2 * pad - R) // st + 1, (W + 2 * pad - R) // st + 1, Ci, Co, R, R)
              for (N, H, W, Ci, Co, R, st, pad) in layers)
    regions = [torch.empty(big, dtype=torch.uint8, device="cuda") for _ in range(2)]
    flip, pend, keep, want, kinds = 0, None, [], [], set()
    for (N, H, W, Ci, Co, R, stride, pad) in layers:
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
        x = torch.randn(G, N, H, W, Ci, generator=g).cuda().to(dtype)
        dy = torch.randn(G, N, Ho, Wo, Co, generator=g).cuda().to(dtype)
        args = (dt, G, N, H, W, Ci, Co, R, R, stride, pad, dy[0].numel(), x[0].numel(), Co * Ci * R * R)
        for acc in (0, 1):
            base = torch.randn(G, Co, Ci, R, R, generator=g).cuda()
            ref, out = base.clone(), base.clone()
            work0 = torch.empty(big, dtype=torch.uint8, device="cuda")
            L.check(lib.ieee_conv2d_wgrad(L.ptr(dy), L.ptr(x), L.ptr(ref), L.ptr(work0), *args, acc, L.stream()))
            d = L.WgradReduceDesc()
            if pend is not None:       # the region the pending reduction reads must not be handed in again
                rc = lib.ieee_conv2d_wgrad_chained(L.ptr(dy), L.ptr(x), L.ptr(out), L.ptr(regions[flip ^ 1]), *args, acc,
                                                   ctypes.addressof(pend), ctypes.addressof(d), L.stream())
                assert rc != 0 and b"alternate" in lib.ieee_last_error()
            L.check(lib.ieee_conv2d_wgrad_chained(L.ptr(dy), L.ptr(x), L.ptr(out), L.ptr(regions[flip]), *args, acc,
                                                  ctypes.addressof(pend) if pend is not None else None, ctypes.addressof(d),
                                                  L.stream()))
            kinds.add(d.kind)
            pend = d if d.kind != 0 else None
            if d.kind != 0:
                flip ^= 1
            keep.append((x, dy, out)); want.append(ref)
    if pend is not None:
        L.check(lib.ieee_wgrad_reduce_pending(ctypes.addressof(pend), G, L.stream()))
    torch.cuda.synchronize()
    assert ({0, 1, 2, 3} if dtype == torch.bfloat16 else {1, 2}) <= kinds, kinds
    for i, ((x, dy, out), ref) in enumerate(zip(keep, want)):
        assert torch.equal(out, ref), "layer %d" % (i // 2)


@pytest.mark.parametrize("Ci,Co,R,H,W,REP", [(64, 256, 1, 16, 8, 1), (128, 128, 3, 16, 8, 1), (256, 64, 1, 12, 10, 1), (64, 256, 1, 16, 8, 8),
                                             (128, 128, 3, 16, 8, 4)])
def test_batchnorm_sums_as_fixed_point_totals_equal_the_partial_sum_path(Ci, Co, R, H, W, REP):
    """ieee_conv_next_bn_totals: the conv / dgrad epilogue adds its per-channel sums to int64 fixed-point totals (no-return
    atomics) instead of writing per-tile partials, and ieee_bn2d_fwd_totals / ieee_bn2d_bwd_totals finalize + apply in one
    launch.  Against the partial-sum path (conv -> ieee_bn2d_fwd / dgrad -> ieee_bn2d_bwd) on the same operands: the totals
    are the integer image of the partial sums (2^24 / 2^40), every output of the BatchNorm passes agrees to a float rounding,
    and the totals are bit-identical over repeated launches (integer adds commute).  REP > 1: row tile t adds to replica
    t % REP of the totals and the BatchNorm passes add the replicas up."""
    from ieee_amd import _lib as L, _ops
    lib = L.require_gpu()
    g = torch.Generator().manual_seed(33)
    G, N = 3, 7
    pad = R // 2
    dt = torch.bfloat16
    M = N * H * W
    x = torch.randn(G, N, H, W, Ci, generator=g).cuda().to(dt)
    w = (torch.randn(G, Co, Ci, R, R, generator=g) * 0.05).cuda()
    wp, wpd = _ops.pack_conv_weight(w, dt, 0), _ops.pack_conv_weight(w, dt, 1)
    rb = lib.ieee_conv2d_fwd_stats_rblocks(N, H, W)
    gam, bet = (torch.rand(G, Co, generator=g) + 0.5).cuda(), torch.randn(G, Co, generator=g).cuda()
    res = torch.randn(G, N, H, W, Co, generator=g).cuda().to(dt)
    flags = torch.zeros(4, dtype=torch.int32, device="cuda")          # the range guard's report words: stay zero on O(1) data

    def forward(use_totals):
        part = torch.zeros(G, 2, Co, rb, device="cuda")
        tot = torch.zeros(REP, G, 2, Co, dtype=torch.int64, device="cuda")
        y = torch.empty(G, N, H, W, Co, device="cuda", dtype=dt)
        a = torch.empty_like(y)
        bits = torch.zeros(y.numel() // 8, dtype=torch.uint8, device="cuda")
        stats = torch.zeros(G, 4, Co, device="cuda")
        rm, rv = torch.zeros(G, Co, device="cuda"), torch.ones(G, Co, device="cuda")
        if use_totals:
            L.check(lib.ieee_conv_next_bn_totals(L.ptr(tot), 2 * Co, REP, L.ptr(flags)))
        L.check(lib.ieee_conv2d_fwd(L.ptr(x), L.ptr(wp), L.ptr(y), L.IEEE_BF16, G, N, H, W, Ci, Co, R, R, 1, pad, x[0].numel(),
                                    wp.stride(0), y[0].numel(), L.ptr(part), L.stream()))
        if use_totals:
            L.check(lib.ieee_bn2d_fwd_totals(L.ptr(y), L.ptr(res), L.ptr(a), L.IEEE_BF16, G, M, Co, M * Co, L.ptr(gam), L.ptr(bet), Co,
                                             L.ptr(rm), L.ptr(rv), Co, L.ptr(stats), L.ptr(tot), REP, 0.1, 1e-5, 1, L.ptr(bits), L.ptr(flags), L.stream()))
        else:
            L.check(lib.ieee_bn2d_fwd(L.ptr(y), L.ptr(res), L.ptr(a), L.IEEE_BF16, G, M, Co, M * Co, L.ptr(gam), L.ptr(bet), Co,
                                      L.ptr(rm), L.ptr(rv), Co, L.ptr(stats), L.ptr(part), 0.1, 1e-5, 1, 1, rb, L.ptr(bits), L.stream()))
        return dict(y=y, a=a, bits=bits, stats=stats, rm=rm, rv=rv, part=part, tot=tot)

    p, t, t2 = forward(False), forward(True), forward(True)
    assert torch.equal(t["tot"], t2["tot"]) and torch.equal(t["a"], t2["a"])             # order-independent: the same bits every time
    assert torch.equal(p["y"], t["y"]) and float(t["part"].abs().max()) == 0.0           # same conv stores; no partials written
    want = torch.round(p["part"].double() * 2.0 ** 24).sum(-1)                           # each tile's float sum, converted, added
    assert torch.equal(t["tot"].sum(0), want.to(torch.int64))
    if REP > 1:   # tile t went to replica t % REP
        per = torch.round(p["part"].double() * 2.0 ** 24)
        for r in range(REP):
            assert torch.equal(t["tot"][r], per[..., r::REP].sum(-1).to(torch.int64))
    torch.testing.assert_close(t["stats"], p["stats"], rtol=2e-6, atol=2e-6)
    torch.testing.assert_close(t["rm"], p["rm"], rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(t["rv"], p["rv"], rtol=2e-6, atol=1e-7)
    assert float((t["a"].float() - p["a"].float()).abs().max()) <= 2.0 ** -6 * float(p["a"].float().abs().max())   # <= one bf16 ulp where sc / sh moved a bit
    assert float((t["a"] != p["a"]).float().mean()) < 2e-3 and float((t["bits"] != p["bits"]).float().mean()) < 2e-3
    # a call that fails its argument checks leaves nothing armed for the next launch of the thread
    stale = torch.zeros(REP, G, 2, Co, dtype=torch.int64, device="cuda")
    L.check(lib.ieee_conv_next_bn_totals(L.ptr(stale), 2 * Co, REP, None))
    assert lib.ieee_conv2d_fwd(None, L.ptr(wp), L.ptr(p["y"]), L.IEEE_BF16, G, N, H, W, Ci, Co, R, R, 1, pad, x[0].numel(), wp.stride(0),
                               p["y"][0].numel(), L.ptr(p["part"]), L.stream()) != 0
    again = forward(False)
    assert torch.equal(again["part"], p["part"]) and int(stale.abs().max()) == 0
    # ---- backward: dgrad with the fused sums of the unit in front (mask recomputed from its y and statistics), then its
    # BatchNorm backward
    dyo = torch.randn(G, N, H, W, Co, generator=g).cuda().to(dt)
    ypre = torch.randn(G, N, H, W, Ci, generator=g).cuda().to(dt)
    pst = torch.randn(G, 4, Ci, generator=g).cuda()
    pst[:, 1] = pst[:, 1].abs() + 0.5
    pgam = (torch.rand(G, Ci, generator=g) + 0.5).cuda()

    def backward(use_totals):
        part = torch.zeros(G, 2, Ci, rb, device="cuda")
        tot = torch.zeros(REP, G, 2, Ci, dtype=torch.int64, device="cuda")
        dx = torch.empty(G, N, H, W, Ci, device="cuda", dtype=dt)
        dyp = torch.empty_like(dx)
        dgam, dbet = torch.zeros(G, Ci, device="cuda"), torch.zeros(G, Ci, device="cuda")
        coef = torch.zeros(G, 3, Ci, device="cuda")
        if use_totals:
            L.check(lib.ieee_conv_next_bn_totals(L.ptr(tot), 2 * Ci, REP, L.ptr(flags)))
        L.check(lib.ieee_conv2d_dgrad(L.ptr(dyo), L.ptr(wpd), L.ptr(dx), None, L.IEEE_BF16, G, N, H, W, Ci, Co, R, R, 1, pad,
                                      dyo[0].numel(), wpd.stride(0), dx[0].numel(), L.ptr(part), L.ptr(ypre), None, L.ptr(pst), 0, 1,
                                      L.stream()))
        if use_totals:
            L.check(lib.ieee_bn2d_bwd_totals(L.ptr(dx), None, L.ptr(ypre), L.ptr(dyp), None, L.IEEE_BF16, G, M, Ci, M * Ci, L.ptr(pgam), Ci,
                                             L.ptr(pst), L.ptr(dgam), L.ptr(dbet), Ci, L.ptr(tot), REP, 1, L.ptr(flags), None, L.stream()))
        else:
            L.check(lib.ieee_bn2d_bwd(L.ptr(dx), None, L.ptr(ypre), L.ptr(dyp), None, L.IEEE_BF16, G, M, Ci, M * Ci, L.ptr(pgam), Ci,
                                      L.ptr(pst), L.ptr(dgam), L.ptr(dbet), Ci, L.ptr(part), L.ptr(coef), 0, 1, rb, L.stream()))
        return dict(dx=dx, dy=dyp, dgam=dgam, dbet=dbet, part=part, tot=tot)

    p, t, t2 = backward(False), backward(True), backward(True)
    assert torch.equal(t["tot"], t2["tot"]) and torch.equal(t["dy"], t2["dy"]) and torch.equal(p["dx"], t["dx"])
    want = torch.round(p["part"].double() * 2.0 ** 40).sum(-1)
    assert torch.equal(t["tot"].sum(0), want.to(torch.int64))
    torch.testing.assert_close(t["dgam"], p["dgam"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(t["dbet"], p["dbet"], rtol=1e-5, atol=1e-5)
    assert float((t["dy"].float() - p["dy"].float()).abs().max()) <= 2.0 ** -6 * float(p["dy"].float().abs().max())
    assert float((t["dy"] != p["dy"]).float().mean()) < 2e-3
    assert flags.tolist() == [0, 0, 0, 0]


@pytest.mark.parametrize("direction", ["forward", "backward"])
@pytest.mark.parametrize("REP", [1, 4])
def test_fixed_point_totals_report_what_leaves_their_range(direction, REP):
    """Range guard of the int64 BatchNorm totals (include/ieee_amd.h, ieee_conv_next_bn_totals): a tile may contribute at most
    2^62 / (row tiles) units, so the total never wraps.  The operands are scaled so that the LARGEST tile sum sits at 0.4x,
    0.9x and 2x that share (forward: sum y^2 against 2.7e11 / tiles; backward: sum g, sum g*y against 4.2e6 / tiles):
      0.4x  nothing reported, every output equals the partial-sum path's (reference: torch's fp32 batch_norm,
            torchreid/models/resnet.py:164-184, has no such range);
      0.9x  still exact -- but the total is beyond half the range: word [2] (forward) / [3] (backward) is set;
      2x    the tile is clamped and word [0] / [1] says so -- the caller must not trust these statistics."""
    from ieee_amd import _lib as L, _ops
    lib = L.require_gpu()
    g = torch.Generator().manual_seed(35)
    G, N, H, W, Ci, Co, R = 3, 7, 16, 8, 64, 128, 1
    dt = torch.bfloat16
    M = N * H * W
    tiles = (M + 127) // 128
    x = torch.randn(G, N, H, W, Ci, generator=g).cuda().to(dt)
    w = (torch.randn(G, Co, Ci, R, R, generator=g) * 0.05).cuda()
    rb = lib.ieee_conv2d_fwd_stats_rblocks(N, H, W)
    gam, bet = (torch.rand(G, Co, generator=g) + 0.5).cuda(), torch.randn(G, Co, generator=g).cuda()

    def forward(wscale, use_totals):
        wp = _ops.pack_conv_weight(w * wscale, dt, 0)
        part = torch.zeros(G, 2, Co, rb, device="cuda")
        tot = torch.zeros(REP, G, 2, Co, dtype=torch.int64, device="cuda")
        flags = torch.zeros(4, dtype=torch.int32, device="cuda")
        y = torch.empty(G, N, H, W, Co, device="cuda", dtype=dt)
        a = torch.empty_like(y)
        stats = torch.zeros(G, 4, Co, device="cuda")
        rm, rv = torch.zeros(G, Co, device="cuda"), torch.ones(G, Co, device="cuda")
        if use_totals:
            L.check(lib.ieee_conv_next_bn_totals(L.ptr(tot), 2 * Co, REP, L.ptr(flags)))
        L.check(lib.ieee_conv2d_fwd(L.ptr(x), L.ptr(wp), L.ptr(y), L.IEEE_BF16, G, N, H, W, Ci, Co, R, R, 1, 0, x[0].numel(),
                                    wp.stride(0), y[0].numel(), L.ptr(part), L.stream()))
        if use_totals:
            L.check(lib.ieee_bn2d_fwd_totals(L.ptr(y), None, L.ptr(a), L.IEEE_BF16, G, M, Co, M * Co, L.ptr(gam), L.ptr(bet), Co,
                                             L.ptr(rm), L.ptr(rv), Co, L.ptr(stats), L.ptr(tot), REP, 0.1, 1e-5, 1, None, L.ptr(flags), L.stream()))
        else:
            L.check(lib.ieee_bn2d_fwd(L.ptr(y), None, L.ptr(a), L.IEEE_BF16, G, M, Co, M * Co, L.ptr(gam), L.ptr(bet), Co,
                                      L.ptr(rm), L.ptr(rv), Co, L.ptr(stats), L.ptr(part), 0.1, 1e-5, 1, 1, rb, None, L.stream()))
        return dict(a=a, stats=stats, rv=rv, part=part, tot=tot, flags=flags.tolist())

    # (backward: positive weights and gradients, so that the tile sums of sum g share a sign and ADD UP over the tiles the way
    # the forward's sum y^2 does -- signed sums cancel and would never reach the "half the range" word)
    wpd = _ops.pack_conv_weight(w.abs(), dt, 1)
    dyo = torch.randn(G, N, H, W, Co, generator=g).abs().cuda().to(dt)
    ypre = torch.randn(G, N, H, W, Ci, generator=g).cuda().to(dt)
    pst = torch.randn(G, 4, Ci, generator=g).cuda()
    pst[:, 1] = pst[:, 1].abs() + 0.5
    pgam = (torch.rand(G, Ci, generator=g) + 0.5).cuda()

    def backward(gscale, use_totals):
        dys = (dyo.float() * gscale).to(dt)
        part = torch.zeros(G, 2, Ci, rb, device="cuda")
        tot = torch.zeros(REP, G, 2, Ci, dtype=torch.int64, device="cuda")
        flags = torch.zeros(4, dtype=torch.int32, device="cuda")
        dx = torch.empty(G, N, H, W, Ci, device="cuda", dtype=dt)
        dyp = torch.empty_like(dx)
        dgam, dbet = torch.zeros(G, Ci, device="cuda"), torch.zeros(G, Ci, device="cuda")
        coef = torch.zeros(G, 3, Ci, device="cuda")
        if use_totals:
            L.check(lib.ieee_conv_next_bn_totals(L.ptr(tot), 2 * Ci, REP, L.ptr(flags)))
        L.check(lib.ieee_conv2d_dgrad(L.ptr(dys), L.ptr(wpd), L.ptr(dx), None, L.IEEE_BF16, G, N, H, W, Ci, Co, R, R, 1, 0,
                                      dys[0].numel(), wpd.stride(0), dx[0].numel(), L.ptr(part), L.ptr(ypre), None, L.ptr(pst), 0, 1,
                                      L.stream()))
        if use_totals:
            L.check(lib.ieee_bn2d_bwd_totals(L.ptr(dx), None, L.ptr(ypre), L.ptr(dyp), None, L.IEEE_BF16, G, M, Ci, M * Ci, L.ptr(pgam), Ci,
                                             L.ptr(pst), L.ptr(dgam), L.ptr(dbet), Ci, L.ptr(tot), REP, 1, L.ptr(flags), None, L.stream()))
        else:
            L.check(lib.ieee_bn2d_bwd(L.ptr(dx), None, L.ptr(ypre), L.ptr(dyp), None, L.IEEE_BF16, G, M, Ci, M * Ci, L.ptr(pgam), Ci,
                                      L.ptr(pst), L.ptr(dgam), L.ptr(dbet), Ci, L.ptr(part), L.ptr(coef), 0, 1, rb, L.stream()))
        return dict(dy=dyp, dgam=dgam, dbet=dbet, part=part, tot=tot, flags=flags.tolist())

    run = forward if direction == "forward" else backward
    share = 2.0 ** 62 / (2.0 ** 24 if direction == "forward" else 2.0 ** 40) / tiles     # a tile's share, in units of the sum
    at_one = float(run(1.0, False)["part"].abs().max())                                     # largest tile sum at scale 1
    power = 0.5 if direction == "forward" else 1.0                                          # sum y^2 is quadratic in the scale
    for target, words in ((0.4, [0, 0, 0, 0]), (0.9, None), (2.0, None)):
        scale = (target * share / at_one) ** power
        p, t = run(scale, False), run(scale, True)
        biggest = float(p["part"].abs().max())
        assert 0.8 * target * share < biggest < 1.25 * target * share, (biggest, target * share)      # bf16 operands: roughly on target
        clamp, half = (0, 2) if direction == "forward" else (1, 3)
        if target < 1.0:
            assert t["flags"][clamp] == 0
            want = torch.round(p["part"].double() * (2.0 ** 24 if direction == "forward" else 2.0 ** 40)).to(torch.int64).sum(-1)
            assert torch.equal(t["tot"].sum(0), want)
            if direction == "forward":
                torch.testing.assert_close(t["stats"], p["stats"], rtol=4e-6, atol=1e-30)
                torch.testing.assert_close(t["rv"], p["rv"], rtol=4e-6, atol=1e-30)
                assert float((t["a"] != p["a"]).float().mean()) < 2e-3
            else:
                torch.testing.assert_close(t["dgam"], p["dgam"], rtol=1e-5, atol=1e-5 * float(p["dgam"].abs().max()))
                torch.testing.assert_close(t["dbet"], p["dbet"], rtol=1e-5, atol=1e-5 * float(p["dbet"].abs().max()))
                assert float((t["dy"] != p["dy"]).float().mean()) < 2e-3
        if words is not None:
            assert t["flags"] == words
        elif target < 1.0:      # exact, but close to the edge: the "beyond half the range" word of this direction
            assert t["flags"][half] == 1 and t["flags"][clamp] == 0
        else:                   # clamped: reported, and the total stayed inside +-2^62 (it did not wrap)
            assert t["flags"][clamp] == 1
            assert int(t["tot"].sum(0).abs().max()) <= 2 ** 62
