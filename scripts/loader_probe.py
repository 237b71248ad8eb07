"""Input pipeline at step rate (reference data path: torchreid/data/datasets/dataset.py:335-351 ->
utils/tools.py:98-119 PIL decode -> data/transforms.py:233-326): a synthetic 3-modal JPEG tree in the RGBNT201 layout,
ieee_amd.data.build_loaders(workers = W) feeding REAL Image3MEngine.forward_backward steps (B = 64, bf16).

  python scripts/loader_probe.py [--workers 4,8,16,32] [--steps 40] [--size 256x128]

Per W: triples/s of the loader alone (decode in the workers + the device-side resize / flip / normalise), triples/s of
loader + train step, and the share of the step loop the GPU step spent waiting for data.  `measure()` is what
`bench.py`'s loader leg calls."""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_tree(root, n_ids=64, per_id=8, size=(256, 128), seed=0, quality=90):
    """<root>/RGBNT201/{train_171,test}/{RGB,NI,TI}/<pid6>_cam<X>_0_<j>.jpg: smooth content + mild noise (a photo-like
    compression ratio; pure noise would decode 2-3x slower than camera images)"""
    from PIL import Image
    rng = np.random.RandomState(seed)
    h, w = size
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    nbytes = 0
    for split, ids in (("train_171", range(n_ids)), ("test", range(n_ids, n_ids + 4))):
        for mod in ("RGB", "NI", "TI"):
            os.makedirs(os.path.join(root, "RGBNT201", split, mod))
        for pid in ids:
            for j in range(per_id if split == "train_171" else 2):
                name = "%06d_cam%d_0_%02d.jpg" % (pid, 1 + j % 4, j)
                for mod in ("RGB", "NI", "TI"):
                    f = rng.rand(3, 3).astype(np.float32)
                    img = np.stack([127 + 100 * np.sin(yy * f[c, 0] / 9 + xx * f[c, 1] / 7 + 6 * f[c, 2]) for c in range(3)], -1)
                    img = np.clip(img + rng.randn(h, w, 3) * 6, 0, 255).astype(np.uint8)
                    path = os.path.join(root, "RGBNT201", split, mod, name)
                    Image.fromarray(img, "RGB").save(path, quality=quality)
                    nbytes += os.path.getsize(path)
    return nbytes


def build_engine(B, device):
    from bench import _FakeDM
    from ieee_amd.engine import Image3MEngine
    from ieee_amd.models import build_model
    from ieee_amd.optim import build_optimizer
    torch.manual_seed(0)
    model = build_model("ieee3modalPart", num_classes=171, loss="margin", pretrained=False, use_gpu=True,
                        compute_dtype=torch.bfloat16, device=device)
    opt = build_optimizer(model, optim="sgd", lr=1e-3, weight_decay=5e-4, momentum=0.9)
    eng = Image3MEngine(_FakeDM(171), model, opt, margin=1, weight_m=1, weight_x=1, use_gpu=True, label_smooth=True)
    eng.defer_summary = True
    model.train()
    return eng


def _cycle(loader):
    while True:
        for batch in loader:
            yield batch


def measure(workers=(4, 8, 16, 32), steps=40, warm=8, B=64, size=(256, 128), engine=None, device=None, root=None, prefetch=None, rounds=3):
    """returns the `loader` object of the bench line"""
    import contextlib
    import io
    from ieee_amd.data import RGBNT201, build_loaders
    device = device or torch.device("cuda", torch.cuda.current_device())
    own_root = root is None
    root = root or tempfile.mkdtemp(prefix="ieee_loader_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    out = {"batch": B, "source_size": "%dx%d" % size, "steps": steps, "per_workers": {}}
    try:
        t0 = time.time()
        # 64 identities x 8 triples: an epoch of the identity sampler is 8 batches of 64; the probe cycles over epochs
        jpeg_bytes = make_tree(root, n_ids=64, per_id=8, size=size)
        out["tree"] = {"jpeg_files": 64 * 8 * 3 + 4 * 2 * 3, "jpeg_MB": jpeg_bytes / 1e6, "build_s": time.time() - t0,
                       "where": root.split("/")[1]}
        with contextlib.redirect_stdout(io.StringIO()):
            ds = RGBNT201(root=root)
        # every file four times under four identity ranges: an epoch of 32 batches instead of 8 without a larger tree (RGBNT201's
        # own epoch is ~60 batches of 64; at every epoch boundary the pipeline drains and refills)
        n_ids = len({r[1] for r in ds.train})
        ds.train = [(r[0], r[1] + n_ids * k, r[2], r[3]) for k in range(4) for r in ds.train]
        eng = engine or build_engine(B, device)
        # the train step alone on a resident batch of the same shape: what the loader has to keep up with
        g = torch.Generator().manual_seed(0)
        res = {"img": [torch.randn(B, 3, 256, 128, generator=g).to(device) for _ in range(3)], "pid": (torch.arange(B) // 4).to(device),
               "camid": torch.zeros(B, dtype=torch.long), "impath": "", "timeid": torch.zeros(B, dtype=torch.long)}
        for _ in range(warm):
            eng.forward_backward(res)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(steps):
            eng.forward_backward(res)
        torch.cuda.synchronize()
        step_ms = (time.time() - t0) / steps * 1e3
        out["resident_step_ms"] = step_ms
        out["host_to_device_MB_per_step"] = B * 3 * size[0] * size[1] * 3 / 1e6        # decoded uint8 HWC, before the resize
        # one loader per worker count, built once and kept (persistent workers); the worker counts are measured in
        # INTERLEAVED rounds -- W1, W2, W3, W1, W2, W3, ... -- so that a drift of the box hits every setting alike, and every
        # setting is reported as median and min - max over the rounds
        loaders, its, alone = {}, {}, {}
        for W in workers:
            with contextlib.redirect_stdout(io.StringIO()):
                kw = {} if prefetch is None else {"prefetch": prefetch}
                loaders[W], _, _ = build_loaders(ds, height=256, width=128, batch_size_train=B, num_instances=4, workers=W, **kw)
            it = its[W] = _cycle(loaders[W])
            for _ in range(warm):          # worker start-up, table upload
                b = next(it)
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(steps):
                b = next(it)
            torch.cuda.synchronize()
            alone[W] = steps * B / (time.time() - t0)
        per_epoch = len(loaders[workers[0]])
        runs = {W: [] for W in workers}
        for r in range(rounds):
            for W in workers:
                it = its[W]
                for _ in range(3):
                    eng.forward_backward(next(it))
                torch.cuda.synchronize()
                t0, waits = time.time(), []
                for _ in range(steps):
                    t1 = time.time()
                    b = next(it)
                    waits.append(time.time() - t1)
                    eng.forward_backward(b)
                torch.cuda.synchronize()
                dt = time.time() - t0
                runs[W].append({"ms_per_step": dt / steps * 1e3, "frac": step_ms / (dt / steps * 1e3),
                                "wait_ms_per_step": sum(waits) / steps * 1e3, "longest_wait_ms": max(waits) * 1e3,
                                "waits_over_2ms": sum(1 for w in waits if w > 2e-3)})
        med = lambda v: sorted(v)[len(v) // 2]
        for W in workers:
            fr = [x["frac"] for x in runs[W]]
            out["per_workers"][str(W)] = {
                "loader_alone_triples_per_s": alone[W],
                "with_train_step_triples_per_s": B / (med([x["ms_per_step"] for x in runs[W]]) * 1e-3),
                "ms_per_step": med([x["ms_per_step"] for x in runs[W]]),
                "frac_of_resident_step_rate": med(fr), "frac_min": min(fr), "frac_max": max(fr),
                "spread_between_rounds": (max(fr) - min(fr)) / med(fr),
                "host_wait_for_batch_ms_per_step": med([x["wait_ms_per_step"] for x in runs[W]]),
                "longest_single_wait_ms": max(x["longest_wait_ms"] for x in runs[W]),
                "waits_over_2ms_per_round": [x["waits_over_2ms"] for x in runs[W]], "rounds": rounds}
        out["batches_per_epoch"] = per_epoch
        del its, loaders
        # the smallest worker count whose WORST round keeps the GPU step at >= 97 % of the resident-batch rate
        ok = [int(w) for w, v in out["per_workers"].items() if v["frac_min"] >= 0.97]
        best = max(out["per_workers"].items(), key=lambda kv: kv[1]["frac_min"])
        out["recommended_workers"] = min(ok) if ok else int(best[0])
        out["gpu_step_stops_waiting_at_workers"] = min(ok) if ok else None
        out["recommendation_rule"] = "smallest worker count whose minimum over %d interleaved rounds is >= 0.97 of the resident-batch rate" % rounds
        out["host_logical_cpus"] = os.cpu_count()
    finally:
        if own_root:
            shutil.rmtree(root, ignore_errors=True)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", default="4,8,16,32")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--size", default="256x128")
    ap.add_argument("--prefetch", type=int, default=None)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    h, w = (int(v) for v in a.size.split("x"))
    print(json.dumps(measure(tuple(int(v) for v in a.workers.split(",")), steps=a.steps, size=(h, w), prefetch=a.prefetch, rounds=a.rounds), indent=1))
