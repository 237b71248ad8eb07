"""exercise the staged data-parallel step (async backward parts, comm stream, RCCL all-reduce of the 13 gradient slices)
on ONE GPU: a 1-rank nccl process group, IEEE_FORCE_DP_PATH=1.  Prints the step time of the staged path next to the
plain single-GPU path (the difference is what the staging itself costs per rank)."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, ".")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
from bench import _FakeDM, make_batch  # noqa: E402
from ieee_amd.engine import Image3MEngine  # noqa: E402
from ieee_amd.models import build_model  # noqa: E402
from ieee_amd.optim import build_optimizer  # noqa: E402


def build_engine_and_batch(B):
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = build_model("ieee3modalPart", num_classes=171, loss="margin", pretrained=False, use_gpu=True,
                        compute_dtype=torch.bfloat16, device=dev)
    opt = build_optimizer(model, optim="sgd", lr=1e-3, weight_decay=5e-4, momentum=0.9)
    eng = Image3MEngine(_FakeDM(171), model, opt, margin=1, weight_m=1, weight_x=1, use_gpu=True, label_smooth=True)
    model.train()
    return eng, make_batch(B, seed=0, device=dev)


for forced in ("0", "1", "1", "0", "1", "0"):
    os.environ["IEEE_FORCE_DP_PATH"] = forced
    eng, batch = build_engine_and_batch(64)
    for _ in range(5):
        eng.forward_backward(batch)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        eng.forward_backward(batch)
    torch.cuda.synchronize()
    print("staged=%s  %.3f ms/step" % (forced, (time.time() - t0) / 20 * 1e3))
    del eng
dist.destroy_process_group()
