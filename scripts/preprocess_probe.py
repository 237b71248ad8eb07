"""throughput of the device-side transform (N2): images/s and GB/s for a batch of 3 x 64 decoded images"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from ieee_amd import _lib  # noqa: E402
from ieee_amd.data import DeviceTransform, resample_tables  # noqa: E402

lib = _lib.require_gpu()
for hs, ws in ((256, 128), (300, 150), (512, 256), (128, 64)):
    n = 192
    src = torch.randint(0, 256, (n, hs, ws, 3), dtype=torch.uint8, device="cuda")
    tr = DeviceTransform(256, 128, "random_flip")
    t, bh, kh, bv, kv = tr._tables_on(src.device, hs, ws)
    dst = torch.empty(n, 3, 256, 128, device="cuda")
    tmp = torch.empty(n, max(t["tmp_rows"], 1), 128, 3, dtype=torch.uint8, device="cuda")
    fl = torch.randint(0, 2, (n,), dtype=torch.uint8, device="cuda")
    mean = (_lib.ctypes.c_float * 3)(0.485, 0.456, 0.406)
    std = (_lib.ctypes.c_float * 3)(0.229, 0.224, 0.225)

    def run():
        _lib.check(lib.ieee_resize_flip_normalize(
            _lib.ptr(src), _lib.ptr(dst), _lib.ptr(tmp), n, hs, ws, 256, 128, _lib.ptr(bh) if t["need_h"] else None,
            _lib.ptr(kh) if t["need_h"] else None, t["ksize_h"], _lib.ptr(bv) if t["need_v"] else None,
            _lib.ptr(kv) if t["need_v"] else None, t["ksize_v"], t["ybox_first"], t["tmp_rows"], _lib.ptr(fl), mean, std,
            _lib.stream()))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(50):
        run()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 50
    byts = src.numel() + dst.numel() * 4 + (2 * tmp.numel() if t["need_h"] else 0)
    print("%dx%d -> 256x128: %.1f us per 192 images, %.2f M images/s, %.0f GB/s algorithmic" % (hs, ws, dt * 1e6, n / dt / 1e6, byts / dt / 1e9))
