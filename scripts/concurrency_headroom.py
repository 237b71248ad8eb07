"""how much throughput is left on the table by the dependency chain of ONE step: run two independent training
processes' worth of work concurrently (two models, two threads, two streams) and compare with one alone"""
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
from bench import _FakeDM, make_batch  # noqa: E402
from ieee_amd.engine import Image3MEngine  # noqa: E402
from ieee_amd.models import build_model  # noqa: E402
from ieee_amd.optim import build_optimizer  # noqa: E402

dev = torch.device("cuda", 0)


def build():
    torch.manual_seed(0)
    model = build_model("ieee3modalPart", num_classes=171, loss="margin", pretrained=False, use_gpu=True,
                        compute_dtype=torch.bfloat16, device=dev)
    opt = build_optimizer(model, optim="sgd", lr=1e-3, weight_decay=5e-4, momentum=0.9)
    eng = Image3MEngine(_FakeDM(171), model, opt, margin=1, weight_m=1, weight_x=1, use_gpu=True, label_smooth=True)
    model.train()
    return eng


def worker(eng, batch, steps, stream, out, i):
    with torch.cuda.stream(stream):
        for _ in range(3):
            eng.forward_backward(batch)
        stream.synchronize()
        t0 = time.time()
        for _ in range(steps):
            eng.forward_backward(batch)
        stream.synchronize()
        out[i] = time.time() - t0


engs = [build(), build()]
batch = make_batch(64, seed=0, device=dev)
K = 20
out = [0, 0]
worker(engs[0], batch, K, torch.cuda.Stream(), out, 0)
print("one engine alone: %.2f ms/step" % (out[0] / K * 1e3))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
th = [threading.Thread(target=worker, args=(engs[i], batch, K, streams[i], out, i)) for i in range(2)]
t0 = time.time()
for t in th:
    t.start()
for t in th:
    t.join()
print("two engines concurrently: %.2f / %.2f ms/step each -> %.2f ms per step-equivalent" % (out[0] / K * 1e3, out[1] / K * 1e3, max(out) / K / 2 * 1e3))
