"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's image transform chain
for the 3-modal datasets (torchreid/data/transforms.py:233-326 with the yaml's transforms=['random_flip']):
    Resize((H, W))  -> PIL Image.resize((W, H), BILINEAR)   [torchvision.transforms.Resize on a PIL image]
    RandomHorizontalFlip(p=0.5)                              [one torch.rand(1) per image, independent per modality:
                                                              dataset.py:341-343 transforms each modality separately]
    ToTensor()      -> uint8 HWC -> float32 CHW / 255
    Normalize(mean, std) -> (x - mean) / std in float32
The resize is Pillow's two-pass 8-bit resampler (third-party dependency of the reference, absent from /root/reference:
Pillow, `src/libImaging/Resample.c`; restated from its published algorithm): per output index a window of
source pixels weighted by a triangle filter whose support grows with the downscale factor (antialiasing), weights
normalised, converted to 22-bit fixed point, accumulated in int32 with a rounding bias and clipped to 8 bits; the
horizontal pass runs first (only over the source rows the vertical pass needs) and writes an 8-bit intermediate.
Pinned by tests/golden/transform_golden.npz, generated from Pillow itself by tests/golden/gen_transform_golden.py."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def precompute_coeffs(in_size, out_size):
    """bounds [out][2] (first source index, count) and fixed-point weights [out][ksize] of one axis (box = whole axis)"""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale                      # bilinear: support 1
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ww = 0.0
        for x in range(xmax):
            w = (x + xmin - center + 0.5) * ss
            w = -w if w < 0 else w
            w = 1.0 - w if w < 1.0 else 0.0
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            kk[xx, :xmax] /= ww
        bounds[xx] = (xmin, xmax)
    fixed = np.where(kk < 0, np.trunc(-0.5 + kk * (1 << PRECISION_BITS)), np.trunc(0.5 + kk * (1 << PRECISION_BITS)))
    return bounds, fixed.astype(np.int32), ksize


def _clip8(acc):
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resample_plan(hs, ws, ho, wo):
    """what ImagingResample decides for an (hs, ws) -> (ho, wo) resize: which passes run and over which rows"""
    need_h, need_v = wo != ws, ho != hs
    bh, kh, ksh = precompute_coeffs(ws, wo)
    bv, kv, ksv = precompute_coeffs(hs, ho)
    ybox_first = int(bv[0, 0])
    ybox_last = int(bv[ho - 1, 0] + bv[ho - 1, 1])
    return dict(need_h=need_h, need_v=need_v, bh=bh, kh=kh, ksize_h=ksh, bv=bv, kv=kv, ksize_v=ksv,
                ybox_first=ybox_first, tmp_rows=ybox_last - ybox_first)


def pil_bilinear_resize_u8(img, ho, wo):
    """img uint8 [hs][ws][c] -> uint8 [ho][wo][c], bit-identical to PIL's Image.resize((wo, ho), BILINEAR)"""
    hs, ws, _ = img.shape
    if (hs, ws) == (ho, wo):
        return img.copy()                           # Image.resize returns self.copy()
    p = resample_plan(hs, ws, ho, wo)
    cur = img
    bv = p["bv"].copy()
    if p["need_h"]:
        rows = cur[p["ybox_first"]:p["ybox_first"] + p["tmp_rows"]].astype(np.int64)
        out = np.empty((p["tmp_rows"], wo, img.shape[2]), dtype=np.uint8)
        for xx in range(wo):
            xmin, cnt = p["bh"][xx]
            acc = (rows[:, xmin:xmin + cnt, :] * p["kh"][xx, :cnt].astype(np.int64)[None, :, None]).sum(1)
            out[:, xx, :] = _clip8(acc + (1 << (PRECISION_BITS - 1)))
        cur = out
        bv[:, 0] -= p["ybox_first"]
    if p["need_v"]:
        src = cur.astype(np.int64)
        out = np.empty((ho, cur.shape[1], img.shape[2]), dtype=np.uint8)
        for yy in range(ho):
            ymin, cnt = bv[yy]
            acc = (src[ymin:ymin + cnt] * p["kv"][yy, :cnt].astype(np.int64)[:, None, None]).sum(0)
            out[yy] = _clip8(acc + (1 << (PRECISION_BITS - 1)))
        cur = out
    return cur


def to_tensor_normalize(img_u8, mean, std, flip=False):
    """uint8 HWC -> float32 CHW: optional horizontal flip, /255, (x - mean) / std, every step in float32"""
    x = img_u8[:, ::-1, :] if flip else img_u8
    t = np.ascontiguousarray(x.transpose(2, 0, 1)).astype(np.float32) / np.float32(255)
    m = np.asarray(mean, dtype=np.float32)[:, None, None]
    s = np.asarray(std, dtype=np.float32)[:, None, None]
    return ((t - m) / s).astype(np.float32)
