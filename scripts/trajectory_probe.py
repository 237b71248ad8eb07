"""prints the loss / accuracy curves of tests/util_trajectory.run_training in the fp32 parity mode, an fp32 control with
jittered starting parameters, and the bf16 speed mode (calibration of tests/test_trajectory_gpu.py's bands)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.util_trajectory import run_training, tamed_state, smooth

kw = dict(epochs=int(os.environ.get("EPOCHS", 12)), batches_per_epoch=10, milestones=(8, 10),
          noise=float(os.environ.get("NOISE", 0.5)), eval_noise={"easy": 0.5, "n4": 4.0, "n6": 6.0, "n8": 8.0, "n10": 10.0, "n14": 14.0},
          lr=float(os.environ.get("LR", 1e-3)))
t0 = time.time()
st = tamed_state(171)
print("state generated in %.1f s" % (time.time() - t0))
runs = {}
for name, dt, pert in (("fp32", torch.float32, 0.0), ("fp32_jit", torch.float32, 2.0 ** -12), ("bf16", torch.bfloat16, 0.0)):
    t0 = time.time()
    r = run_training(dt, st, perturb=pert, **kw)
    r.pop("model"); r.pop("engine")
    runs[name] = r
    torch.cuda.empty_cache()
    print("%-9s %.1f s  loss %8.3f -> %8.3f  acc %5.1f -> %5.1f  LossM %.3f -> %.3f  mAP %.4f rank1 %.4f lr_end %.1e" % (
        name, time.time() - t0, r["loss"][0], r["loss"][-5:].mean(), r["acc"][:5].mean(), r["acc"][-5:].mean(),
        r["LossM"][0], r["LossM"][-5:].mean(), r["mAP"], r["rank1"], r["lr_end"]), r["evals"])
ref = runs["fp32"]
drop = ref["loss"][0] - ref["loss"][-5:].mean()
for name in ("fp32_jit", "bf16"):
    d = np.abs(smooth(runs[name]["loss"]) - smooth(ref["loss"]))
    print("%-9s max |smoothed loss - fp32| = %.3f (%.2f %% of the drop %.2f), at step %d; acc diff max %.2f" % (
        name, d.max(), 100 * d.max() / drop, drop, int(d.argmax()), np.abs(smooth(runs[name]["acc"]) - smooth(ref["acc"])).max()))
for i in range(0, len(ref["loss"]), 5):
    print("step %3d  fp32 %8.3f  jit %8.3f  bf16 %8.3f | acc %5.1f %5.1f %5.1f" % (
        i, ref["loss"][i], runs["fp32_jit"]["loss"][i], runs["bf16"]["loss"][i], ref["acc"][i], runs["fp32_jit"]["acc"][i], runs["bf16"]["acc"][i]))
