"""how long do small pageable host->device copies take before / after a B = 64 engine step has run in the process?"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import _FakeDM, make_batch


def timeit(label):
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        x = torch.from_numpy(np.arange(256, dtype=np.int32)).to("cuda")
    torch.cuda.synchronize()
    a = (time.time() - t0) / 20 * 1e3
    pin = torch.arange(256, dtype=torch.int32).pin_memory()
    t0 = time.time()
    for _ in range(20):
        y = pin.to("cuda", non_blocking=True)
    torch.cuda.synchronize()
    b = (time.time() - t0) / 20 * 1e3
    t0 = time.time()
    for _ in range(20):
        z = torch.empty(1 << 20, device="cuda")
    c = (time.time() - t0) / 20 * 1e3
    print("%-34s pageable 1 KB copy %.3f ms | pinned async %.3f ms | 4 MB device alloc %.3f ms" % (label, a, b, c), flush=True)


torch.zeros(1).cuda()
timeit("fresh process")
from ieee_amd.engine import Image3MEngine
from ieee_amd.models import build_model
from ieee_amd.optim import build_optimizer
dev = torch.device("cuda", 0)
model = build_model("ieee3modalPart", num_classes=171, loss="margin", pretrained=False, use_gpu=True, compute_dtype=torch.bfloat16, device=dev)
timeit("model built")
eng = Image3MEngine(_FakeDM(171), model, build_optimizer(model, optim="sgd", lr=1e-3, weight_decay=5e-4, momentum=0.9), margin=1, use_gpu=True)
model.train()
batch = make_batch(64, 0, dev)
for _ in range(3):
    eng.forward_backward(batch)
torch.cuda.synchronize()
timeit("after 3 engine steps (alive)")
del eng, model, batch
gc.collect()
timeit("engine + model deleted")
torch.cuda.empty_cache()
timeit("cache emptied")


def fork_once(hold=0.0):
    pid = os.fork()
    if pid == 0:
        time.sleep(hold)
        os._exit(0)
    return pid


print("--- now with forks")
model = build_model("ieee3modalPart", num_classes=171, loss="margin", pretrained=False, use_gpu=True, compute_dtype=torch.bfloat16, device=dev)
eng = Image3MEngine(_FakeDM(171), model, build_optimizer(model, optim="sgd", lr=1e-3, weight_decay=5e-4, momentum=0.9), margin=1, use_gpu=True)
model.train()
batch = make_batch(64, 0, dev)
for _ in range(3):
    eng.forward_backward(batch)
torch.cuda.synchronize()
t0 = time.time(); p = fork_once(); os.waitpid(p, 0); print("fork+exit %.3f s" % (time.time() - t0))
timeit("after fork (child gone)")
t0 = time.time(); p = fork_once(3.0); print("fork %.3f s" % (time.time() - t0))
timeit("child alive, 1st")
timeit("child alive, 2nd")
os.waitpid(p, 0)
timeit("child reaped")
s = eng.forward_backward(batch); torch.cuda.synchronize()
t0 = time.time()
for _ in range(5):
    eng.forward_backward(batch)
torch.cuda.synchronize()
print("5 engine steps after the forks: %.2f ms each" % ((time.time() - t0) / 5 * 1e3))
p = fork_once(3.0)
t0 = time.time()
for _ in range(5):
    eng.forward_backward(batch)
torch.cuda.synchronize()
print("5 engine steps with a forked child alive: %.2f ms each" % ((time.time() - t0) / 5 * 1e3))
os.waitpid(p, 0)
