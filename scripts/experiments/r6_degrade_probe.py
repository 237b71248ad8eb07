"""why does a model degraded to the partial-sum BatchNorm path by the range guard give another loss than one that never used
the fixed-point totals?  (tests/test_trajectory_gpu.py::test_engine_degrades...)"""
import os
import sys
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_trajectory_gpu import tamed_state  # noqa: E402
from tests.util_trajectory import _DM, make_train_set  # noqa: E402
from ieee_amd.engine import Image3MEngine  # noqa: E402
from ieee_amd.models import build_model  # noqa: E402
from ieee_amd.optim import build_optimizer  # noqa: E402

warnings.simplefilter("always")


def fresh(dtype=torch.bfloat16):
    st = {k: v.clone() for k, v in tamed_state(171).items()}
    m = build_model("ieee3modalPart", num_classes=171, loss="margin", pretrained=False, compute_dtype=dtype)
    m.load_state_dict(st)
    m.train()
    eng = Image3MEngine(_DM(171, [], {}), m, build_optimizer(m, optim="sgd", lr=0.0, weight_decay=0.0, momentum=0.0), margin=1, use_gpu=True)
    return m, eng


xs, pids, cams = make_train_set(2, 4, 21, 0.5)
batch = {"img": xs, "pid": pids, "camid": cams, "impath": "", "timeid": pids * 0}


def blow(m):
    with torch.no_grad():
        dict(m._param_items)["backbone.1.layer1.0.conv2.weight"].mul_(1e5)


def run(tag, pre_steps, off_before, dtype=torch.bfloat16):
    m, eng = fresh(dtype)
    if off_before:
        m._bn_totals_off = True
    p0 = m._flat_params.clone()
    for _ in range(pre_steps):
        eng.forward_backward(batch)
    blow(m)
    p1 = m._flat_params.clone()
    losses = [float(eng.forward_backward(batch)["loss"]) for _ in range(4)]
    torch.cuda.synchronize()
    print("%-46s" % tag, ["%.6f" % l for l in losses], "params moved by steps:", int((m._flat_params != p1).sum()),
          "nan:", int(torch.isnan(m._flat_params).sum()), "buffers nan:", int(torch.isnan(m._flat_buffers).sum()))


run("bf16, totals on (degrades at step 1 read)", 0, False)
run("bf16, 1 healthy step, totals on", 1, False)
run("bf16, totals off from the start", 0, True)
run("bf16, 1 healthy step, totals off from start", 1, True)
run("fp32", 0, False, torch.float32)
run("fp32, 1 healthy step", 1, False, torch.float32)
