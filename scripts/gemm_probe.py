"""micro-benchmark of single conv / distmat launches (kernel tuning aid; not part of the product)"""
import sys
import time

import torch

sys.path.insert(0, ".")
from ieee_amd import _ops  # noqa: E402


def timeit(f, reps=20):
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dt = torch.bfloat16
    B = 64
    shapes = [(2048, 2048, 1, 1, 16, 8), (512, 512, 3, 1, 16, 8), (512, 2048, 1, 1, 16, 8), (256, 256, 3, 1, 16, 8),
              (64, 256, 1, 1, 64, 32), (256, 64, 1, 1, 64, 32), (64, 64, 3, 1, 64, 32), (128, 128, 3, 1, 32, 16),
              (1024, 256, 1, 1, 16, 8), (128, 512, 1, 1, 32, 16)]
    which = sys.argv[1:] or ["fwd", "dgrad", "wgrad"]
    for Ci, Co, R, st, H, W in shapes:
        x = torch.randn(3, B, H, W, Ci, device="cuda").to(dt)
        w = torch.randn(3, Co, Ci, R, R, device="cuda") * 0.05
        wp, wpd = _ops.pack_conv_weight(w, dt, 0), _ops.pack_conv_weight(w, dt, 1)
        pad = R // 2
        y = _ops.conv2d_fwd(x, wp, Co, R, R, st, pad)
        fl = 3 * 2.0 * B * H * W * Co * Ci * R * R
        out = "%4d->%4d k%d %2dx%2d :" % (Ci, Co, R, H, W)
        if "fwd" in which:
            us = timeit(lambda: _ops.conv2d_fwd(x, wp, Co, R, R, st, pad))
            out += "  fwd %7.1f us %6.0f TF" % (us, fl / us / 1e6)
        if "fwdstats" in which:
            from ieee_amd import _lib as L
            lib = L.load()
            rb = lib.ieee_conv2d_fwd_stats_rblocks(B, H // st, W // st)
            part = torch.zeros(3, rb, 2, Co, device="cuda")
            yy = torch.empty_like(y)

            def f():
                L.check(lib.ieee_conv2d_fwd(L.ptr(x), L.ptr(wp), L.ptr(yy), L.IEEE_BF16, 3, B, H, W, Ci, Co, R, R, st, pad,
                                            x[0].numel(), wp.stride(0), yy[0].numel(), L.ptr(part), L.stream()))
            us = timeit(f)
            out += "  fwd+stats %7.1f us" % us
        if "dgrad" in which:
            us = timeit(lambda: _ops.conv2d_dgrad(y, wpd, (H, W), Ci, R, R, st, pad))
            out += "  dgrad %7.1f us %6.0f TF" % (us, fl / us / 1e6)
        if "wgrad" in which:
            dw = torch.zeros(3, Co, Ci, R, R, device="cuda")
            us = timeit(lambda: _ops.conv2d_wgrad(y, x, R, R, st, pad, out=dw))
            out += "  wgrad %7.1f us %6.0f TF" % (us, fl / us / 1e6)
        print(out)


if __name__ == "__main__":
    main()
