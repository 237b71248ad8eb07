"""how long does the HOST take to enqueue one train step (B = 64, bf16)?  Three steps are enqueued behind a device-wide
synchronisation without waiting for the device (the deferred-summary ring lets the host run 3 steps ahead); the device
then needs ~15 ms per step to execute them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import _FakeDM, make_batch
from ieee_amd.engine import Image3MEngine
from ieee_amd.models import build_model
from ieee_amd.optim import build_optimizer
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = build_model("ieee3modalPart", num_classes=171, loss="margin", pretrained=False, use_gpu=True, compute_dtype=torch.bfloat16, device=dev)
eng = Image3MEngine(_FakeDM(171), model, build_optimizer(model, optim="sgd", lr=1e-3, weight_decay=5e-4, momentum=0.9), margin=1,
                    weight_m=1, weight_x=1, use_gpu=True, label_smooth=True)
eng.defer_summary = True; eng.resident_batch = True
model.train()
batch = make_batch(64, 0, dev)
for _ in range(10):
    eng.forward_backward(batch)
torch.cuda.synchronize()
host, total = [], []
for rep in range(8):
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        eng.forward_backward(batch)
    t1 = time.time()
    torch.cuda.synchronize()
    t2 = time.time()
    host.append((t1 - t0) / 3 * 1e3); total.append((t2 - t0) / 3 * 1e3)
print("host enqueue per step: %s ms (median %.2f); enqueue + execute per step: median %.2f ms" % (
    ["%.2f" % h for h in host], sorted(host)[len(host) // 2], sorted(total)[len(total) // 2]))
