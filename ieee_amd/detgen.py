"""Deterministic, platform-independent tensor generator.

Weights (109.5 M parameters) cannot be committed as fixtures,
so parity tests regenerate them identically in the build container (where the
reference is imported to produce golden vectors) and on the GPU box.  Every
value is a pure function of (tensor name, flat index, seed): a splitmix64 hash
mapped to [0,1) and, for "normal" draws, an Irwin-Hall sum of four uniforms.
Only IEEE-exact operations (integer ops, add, multiply by a constant) are used,
so the result does not depend on libm / SIMD code paths.
"""
from __future__ import annotations

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        z = z ^ (z >> np.uint64(31))
    return z


def det_uniform(name: str, n: int, seed: int = 0, stream: int = 0) -> np.ndarray:
    """n float64 values in [0, 1), pure function of (name, seed, stream, index)."""
    base = (_fnv1a64(name) ^ ((seed * 0xD6E8FEB86659FD93) & 0xFFFFFFFFFFFFFFFF)
            ^ ((stream * 0xA0761D6478BD642F) & 0xFFFFFFFFFFFFFFFF))
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _splitmix64((idx * np.uint64(0x2545F4914F6CDD1D) + np.uint64(base)) & _MASK)
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def det_normal(name: str, n: int, seed: int = 0) -> np.ndarray:
    """Approximately N(0,1) (Irwin-Hall, 4 uniforms), float64, exact arithmetic only."""
    s = det_uniform(name, n, seed, 0)
    for k in (1, 2, 3):
        s = s + det_uniform(name, n, seed, k)
    return (s - 2.0) * 1.7320508075688772  # var of the sum is 4/12


def det_tensor(name, shape, seed=0, kind="normal", scale=1.0, shift=0.0, dtype=np.float32):
    n = int(np.prod(shape)) if len(shape) else 1
    v = det_normal(name, n, seed) if kind == "normal" else det_uniform(name, n, seed) * 2.0 - 1.0
    return (v * scale + shift).astype(dtype).reshape(shape)


def generate_state(shapes: dict, seed: int = 0, rem_param: float = 0.1) -> dict:
    """Trained-like, well-conditioned values for every state_dict entry.

    `shapes` maps the reference's state_dict keys to shapes.
    conv: N(0, 2/fan_out) (the trunk's kaiming fan_out rule, reference
    torchreid/models/resnet.py:603-620); linear: U(+-1/sqrt(fan_in)); BN gamma
    1+-0.1, beta +-0.1, running_mean +-0.1, running_var in [0.8, 1.2];
    REM.param non-zero so the REM branch is exercised.
    """
    out = {}
    for k, shp in shapes.items():
        shp = tuple(shp)
        leaf = k.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[k] = np.zeros((), dtype=np.int64)
        elif leaf == "param":
            out[k] = np.full(shp, rem_param, dtype=np.float32)
        elif leaf == "running_mean":
            out[k] = det_tensor(k, shp, seed, "uniform", 0.1)
        elif leaf == "running_var":
            out[k] = det_tensor(k, shp, seed, "uniform", 0.2, 1.0)
        elif len(shp) == 4:
            fan_out = shp[0] * shp[2] * shp[3]
            out[k] = det_tensor(k, shp, seed, "normal", float(np.sqrt(2.0 / fan_out)))
        elif len(shp) == 2:
            out[k] = det_tensor(k, shp, seed, "uniform", float(1.0 / np.sqrt(shp[1])))
        elif leaf == "weight":      # BN gamma (1-D)
            out[k] = det_tensor(k, shp, seed, "uniform", 0.1, 1.0)
        elif leaf == "bias":
            out[k] = det_tensor(k, shp, seed, "uniform", 0.1)
        else:
            raise KeyError(k)
    return out


def generate_images(batch: int, seed: int = 0, height: int = 256, width: int = 128):
    """Three [B,3,H,W] float32 ~N(0,1) tensors (what Normalize emits; reference
    torchreid/data/transforms.py:269-272), order [RGB, NI, TI]."""
    return [det_tensor("img.%s" % m, (batch, 3, height, width), seed, "normal")
            for m in ("RGB", "NI", "TI")]


def generate_identity_images(pids, cams, seed: int = 0, height: int = 256, width: int = 128, noise: float = 0.25):
    """Three [n,3,H,W] float32 tensors whose content depends on the identity: per (identity, modality, channel) a
    constant level and a vertical ramp, a small per-camera level, plus N(0, noise^2) pixel noise.  Gives the evaluation
    tests query / gallery sets whose descriptors separate by identity (pure-noise images all pool to nearly the same
    descriptor, and their distances are rounding noise).  Exact arithmetic only, like everything in this module."""
    pids = [int(p) for p in pids]
    cams = [int(c) for c in cams]
    n = len(pids)
    base = generate_images(n, seed, height, width)
    ramp = ((np.arange(height, dtype=np.float64) / float(height - 1)) - 0.5) * 2.0          # -1 .. 1 down the image
    out = []
    for m, x in enumerate(base):
        x = x.astype(np.float64) * noise
        for i in range(n):
            lv = det_uniform("id.level.%d" % pids[i], 9, seed=0)[3 * m:3 * m + 3] * 2.0 - 1.0
            sl = det_uniform("id.slope.%d" % pids[i], 9, seed=0)[3 * m:3 * m + 3] * 2.0 - 1.0
            cl = det_uniform("cam.level.%d" % cams[i], 9, seed=0)[3 * m:3 * m + 3] * 0.2 - 0.1
            for c in range(3):
                x[i, c] += (lv[c] + cl[c]) + sl[c] * ramp[:, None]
        out.append(x.astype(np.float32))
    return out
