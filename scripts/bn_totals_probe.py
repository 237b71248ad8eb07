"""BatchNorm from fixed-point totals against the partial-sum path, launch by launch (GPU box):
  python scripts/bn_totals_probe.py
Per shape: the conv with per-tile partials / with atomics into the totals, and the BatchNorm forward that follows each.
PROBE_REPLICAS=R: the totals in R copies (row tile t adds to copy t % R)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ieee_amd import _lib as L, _ops

lib = L.require_gpu()
dt = torch.bfloat16


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (N, H, W, Ci, Co, R) in ((64, 8, 4, 2048, 512, 1), (64, 16, 8, 1024, 256, 1), (64, 16, 8, 256, 1024, 1), (64, 32, 16, 512, 128, 1), (64, 64, 32, 256, 64, 1)):
    G, pad, M = 3, R // 2, N * H * W
    g = torch.Generator().manual_seed(1)
    x = torch.randn(G, N, H, W, Ci, generator=g).cuda().to(dt)
    w = (torch.randn(G, Co, Ci, R, R, generator=g) * 0.05).cuda()
    wp = _ops.pack_conv_weight(w, dt, 0)
    rb = lib.ieee_conv2d_fwd_stats_rblocks(N, H, W)
    gam, bet = torch.ones(G, Co).cuda(), torch.zeros(G, Co).cuda()
    part = torch.zeros(G, 2, Co, rb, device="cuda")
    REP = int(os.environ.get("PROBE_REPLICAS", "1"))
    tot = torch.zeros(REP, G, 2, Co, dtype=torch.int64, device="cuda")
    y = torch.empty(G, N, H, W, Co, device="cuda", dtype=dt)
    a = torch.empty_like(y)
    stats = torch.zeros(G, 4, Co, device="cuda")
    rm, rv = torch.zeros(G, Co, device="cuda"), torch.ones(G, Co, device="cuda")

    def conv(use_totals):
        if use_totals:
            L.check(lib.ieee_conv_next_bn_totals(L.ptr(tot), 2 * Co, REP, None))
        L.check(lib.ieee_conv2d_fwd(L.ptr(x), L.ptr(wp), L.ptr(y), L.IEEE_BF16, G, N, H, W, Ci, Co, R, R, 1, pad, x[0].numel(),
                                    wp.stride(0), y[0].numel(), L.ptr(part), L.stream()))

    def bn(use_totals):
        if use_totals:
            L.check(lib.ieee_bn2d_fwd_totals(L.ptr(y), None, L.ptr(a), L.IEEE_BF16, G, M, Co, M * Co, L.ptr(gam), L.ptr(bet), Co,
                                             L.ptr(rm), L.ptr(rv), Co, L.ptr(stats), L.ptr(tot), REP, 0.1, 1e-5, 1, None, None, L.stream()))
        else:
            L.check(lib.ieee_bn2d_fwd(L.ptr(y), None, L.ptr(a), L.IEEE_BF16, G, M, Co, M * Co, L.ptr(gam), L.ptr(bet), Co,
                                      L.ptr(rm), L.ptr(rv), Co, L.ptr(stats), L.ptr(part), 0.1, 1e-5, 1, 1, rb, None, L.stream()))

    tot.zero_(); conv(True); torch.cuda.synchronize()
    print("M=%d C=%d->%d tiles=%d: conv partial %.1f us, conv totals %.1f us | bn partial(finalize+apply) %.1f us, bn totals %.1f us | pair partial %.1f, pair totals %.1f"
          % (M, Ci, Co, rb, timed(lambda: conv(False)), timed(lambda: conv(True)), timed(lambda: bn(False)), timed(lambda: bn(True)),
             timed(lambda: (conv(False), bn(False))), timed(lambda: (tot.zero_(), conv(True), bn(True)))), flush=True)
