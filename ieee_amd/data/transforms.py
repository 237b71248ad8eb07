"""Device-side image transforms (reference torchreid/data/transforms.py:233-326 with the live option set of
configs/RGBNT_ieee_part_margin.yaml: Resize, RandomHorizontalFlip, ToTensor, Normalize).  The host only computes
Pillow's per-axis resampling tables (a few hundred integers per distinct source size, cached) and draws the flip
decisions; the pixels are resized, flipped, scaled and normalised by ieee_resize_flip_normalize on the GPU, bit-exactly
what torchvision.transforms does through Pillow on the CPU."""
import math

import numpy as np
import torch

from .. import _lib

PRECISION_BITS = 32 - 8 - 2       # Pillow's 8-bit resampler keeps 22 fractional bits
IMAGENET_MEAN = [0.485, 0.456, 0.406]
IMAGENET_STD = [0.229, 0.224, 0.225]


def _axis_tables(in_size, out_size):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle) filter over a whole axis:
    bounds int32 [out][2] = (first source index, count), weights int32 [out][ksize] (22-bit fixed point)"""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size     # the box edges are C floats
    filterscale = scale if scale > 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    inv = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        x = np.arange(xmax, dtype=np.float64)
        w = np.abs((x + xmin - center + 0.5) * inv)
        w = np.where(w < 1.0, 1.0 - w, 0.0)
        ww = 0.0
        for v in w:                      # same left-to-right double accumulation as the C loop
            ww += float(v)
        if ww != 0.0:
            w = w / ww
        kk[xx, :xmax] = w
        bounds[xx] = (xmin, xmax)
    fixed = np.trunc(np.where(kk < 0, -0.5, 0.5) + kk * (1 << PRECISION_BITS)).astype(np.int32)
    return bounds, fixed, ksize


_TABLE_CACHE = {}


def resample_tables(hs, ws, ho, wo):
    """everything ieee_resize_flip_normalize needs for an (hs, ws) -> (ho, wo) resize, as Pillow's ImagingResample
    plans it: the horizontal pass runs first and only over the source rows the vertical pass will read"""
    key = (hs, ws, ho, wo)
    if key not in _TABLE_CACHE:
        bh, kh, ksh = _axis_tables(ws, wo)
        bv, kv, ksv = _axis_tables(hs, ho)
        y0 = int(bv[0, 0])
        _TABLE_CACHE[key] = dict(need_h=wo != ws, need_v=ho != hs, bounds_h=bh, kk_h=kh, ksize_h=ksh, bounds_v=bv,
                                 kk_v=kv, ksize_v=ksv, ybox_first=y0, tmp_rows=int(bv[ho - 1, 0] + bv[ho - 1, 1]) - y0)
    return _TABLE_CACHE[key]


class DeviceTransform(object):
    """callable(list of uint8 HxWx3 arrays) -> float32 [N,3,height,width] CUDA tensor.  `train` enables the random
    flip when 'random_flip' is among `transforms`; one torch.rand(1) is drawn per image, in call order, exactly like
    torchvision.transforms.RandomHorizontalFlip inside the reference's per-image Compose."""

    SUPPORTED = ('random_flip',)

    def __init__(self, height, width, transforms='random_flip', norm_mean=None, norm_std=None, train=True, device=None):
        if transforms is None:
            transforms = []
        if isinstance(transforms, str):
            transforms = [transforms]
        if not isinstance(transforms, list):
            raise ValueError('transforms must be a list of strings, but found to be {}'.format(type(transforms)))
        transforms = [t.lower() for t in transforms]
        for t in transforms:
            if t not in self.SUPPORTED:
                raise NotImplementedError("transform '%s' is not live in the reference's config (only random_flip) "
                                          "and is not built" % t)
        self.height, self.width = int(height), int(width)
        self.flip = train and 'random_flip' in transforms
        self.mean = np.asarray(IMAGENET_MEAN if norm_mean is None or norm_std is None else norm_mean, dtype=np.float32)
        self.std = np.asarray(IMAGENET_STD if norm_mean is None or norm_std is None else norm_std, dtype=np.float32)
        self.device = device
        self._dev_tables = {}

    def _tables_on(self, dev, hs, ws):
        key = (str(dev), hs, ws)
        if key not in self._dev_tables:
            t = resample_tables(hs, ws, self.height, self.width)
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            self._dev_tables[key] = (t, up(t["bounds_h"]), up(t["kk_h"]), up(t["bounds_v"]), up(t["kk_v"]))
        return self._dev_tables[key]

    _vector_draw_ok = None     # does torch.rand(n) consume the generator like n x torch.rand(1) on this build?  (checked once)

    def draw_flips(self, n):
        """one torch.rand(1) < 0.5 per image, in call order (torchvision's RandomHorizontalFlip inside the reference's per-image
        Compose).  Drawn as ONE torch.rand(n) when this torch build produces the same numbers that way (checked once on a
        saved generator state): 192 interpreter round trips per batch otherwise, under the lock the train step needs too."""
        if not self.flip:
            return np.zeros(n, dtype=np.uint8)
        cls = DeviceTransform
        if cls._vector_draw_ok is None:
            state = torch.get_rng_state()
            a = torch.rand(37)
            torch.set_rng_state(state)
            b = torch.cat([torch.rand(1) for _ in range(37)])
            torch.set_rng_state(state)
            cls._vector_draw_ok = bool(torch.equal(a, b))
        if cls._vector_draw_ok and n > 0:
            return (torch.rand(n) < 0.5).to(torch.uint8).numpy()
        return np.asarray([1 if float(torch.rand(1)) < 0.5 else 0 for _ in range(n)], dtype=np.uint8)

    def __call__(self, images, flips=None):
        lib = _lib.require_gpu()
        dev = torch.device(self.device) if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        n = len(images)
        out = torch.empty((n, 3, self.height, self.width), dtype=torch.float32, device=dev)
        if n == 0:
            return out
        flips = self.draw_flips(n) if flips is None else np.asarray(flips, dtype=np.uint8)
        if torch.is_tensor(images):
            # a whole batch of same-size images already stacked [N, H, W, 3] uint8 by the loader's workers (pinned when the
            # DataLoader pins): one asynchronous copy, no per-image work on this thread
            if images.dim() != 4 or images.shape[3] != 3 or images.dtype != torch.uint8:
                raise ValueError("expected a uint8 [N, H, W, 3] batch, got %s %s" % (images.dtype, tuple(images.shape)))
            hs, ws = int(images.shape[1]), int(images.shape[2])
            t, bh, kh, bv, kv = self._tables_on(dev, hs, ws)
            src = images.to(dev, non_blocking=True)
            # the flip flags through a small ring of pinned buffers: a pageable source would make this copy synchronous --
            # the prefetch thread would stall until its stream has drained behind the train step's kernels
            # (a slot is rewritten 8 calls later -- under three batches; the copy out of it may still be queued behind other
            # streams' work then, e.g. with one hardware queue per priority in a data-parallel job: an event per slot, recorded
            # behind the copy and waited for before the host writes the slot again, makes the reuse safe whatever the backlog)
            ring = self.__dict__.setdefault("_flip_ring", {"at": 0, "bufs": [None] * 8, "done": [None] * 8})
            slot = ring["at"] = (ring["at"] + 1) % 8
            if ring["done"][slot] is not None:
                ring["done"][slot].synchronize()
            if ring["bufs"][slot] is None or ring["bufs"][slot].numel() < n:
                ring["bufs"][slot] = torch.empty(max(n, 256), dtype=torch.uint8).pin_memory()
            ring["bufs"][slot][:n].copy_(torch.from_numpy(flips))
            fl = ring["bufs"][slot][:n].to(dev, non_blocking=True)
            if ring["done"][slot] is None:
                ring["done"][slot] = torch.cuda.Event()
            ring["done"][slot].record(torch.cuda.current_stream(dev))
            tmp = torch.empty((n, t["tmp_rows"], self.width, 3), dtype=torch.uint8, device=dev) if t["need_h"] else None
            mean = (_lib.ctypes.c_float * 3)(*self.mean.tolist())
            std = (_lib.ctypes.c_float * 3)(*self.std.tolist())
            _lib.check(lib.ieee_resize_flip_normalize(
                _lib.ptr(src), _lib.ptr(out), _lib.ptr(tmp) if tmp is not None else None, n, hs, ws, self.height,
                self.width, _lib.ptr(bh) if t["need_h"] else None, _lib.ptr(kh) if t["need_h"] else None, t["ksize_h"],
                _lib.ptr(bv) if t["need_v"] else None, _lib.ptr(kv) if t["need_v"] else None, t["ksize_v"],
                t["ybox_first"], t["tmp_rows"], _lib.ptr(fl), mean, std, _lib.stream()))
            return out
        groups = {}
        for i, im in enumerate(images):
            im = np.asarray(im)
            if im.ndim != 3 or im.shape[2] != 3 or im.dtype != np.uint8:
                raise ValueError("expected uint8 HxWx3 images (PIL 'RGB'), got %s %s" % (im.dtype, im.shape))
            groups.setdefault(im.shape[:2], []).append(i)
        mean = (_lib.ctypes.c_float * 3)(*self.mean.tolist())
        std = (_lib.ctypes.c_float * 3)(*self.std.tolist())
        for (hs, ws), idx in groups.items():
            t, bh, kh, bv, kv = self._tables_on(dev, hs, ws)
            src = torch.from_numpy(np.stack([np.ascontiguousarray(images[i]) for i in idx])).to(dev, non_blocking=True)
            fl = torch.from_numpy(flips[idx].copy()).to(dev)
            dst = out if len(groups) == 1 else torch.empty((len(idx), 3, self.height, self.width), dtype=torch.float32, device=dev)
            tmp = torch.empty((len(idx), t["tmp_rows"], self.width, 3), dtype=torch.uint8, device=dev) if t["need_h"] else None
            _lib.check(lib.ieee_resize_flip_normalize(
                _lib.ptr(src), _lib.ptr(dst), _lib.ptr(tmp) if tmp is not None else None, len(idx), hs, ws, self.height,
                self.width, _lib.ptr(bh) if t["need_h"] else None, _lib.ptr(kh) if t["need_h"] else None, t["ksize_h"],
                _lib.ptr(bv) if t["need_v"] else None, _lib.ptr(kv) if t["need_v"] else None, t["ksize_v"],
                t["ybox_first"], t["tmp_rows"], _lib.ptr(fl), mean, std, _lib.stream()))
            if dst is not out:
                out[torch.as_tensor(idx, device=dev)] = dst
        return out


def build_transforms(height, width, transforms='random_flip', norm_mean=None, norm_std=None, **kwargs):
    """reference transforms.py:233-326: returns (train transform, test transform)"""
    print('Building train transforms ...')
    print('+ resize to {}x{}'.format(height, width))
    tr = DeviceTransform(height, width, transforms, norm_mean, norm_std, train=True)
    if tr.flip:
        print('+ random flip')
    print('+ to torch tensor of range [0, 1]')
    print('+ normalization (mean={}, std={})'.format(tr.mean.tolist(), tr.std.tolist()))
    print('Building test transforms ...')
    print('+ resize to {}x{}'.format(height, width))
    print('+ to torch tensor of range [0, 1]')
    print('+ normalization (mean={}, std={})'.format(tr.mean.tolist(), tr.std.tolist()))
    return tr, DeviceTransform(height, width, [], norm_mean, norm_std, train=False)
