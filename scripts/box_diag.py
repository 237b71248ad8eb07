"""what kind of box is this?  The same B = 64 bf16 train step through the plain single-GPU path, the plain path with the
optimizer on the launch stream (IEEE_OPT_OVERLAP=0), the staged data-parallel path over a 1-rank RCCL group, and without the
weight-gradient stream -- interleaved, with rocm-smi clocks / power sampled while the step runs."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from bench import _FakeDM, make_batch
from ieee_amd.engine import Image3MEngine
from ieee_amd.models import build_model
from ieee_amd.optim import build_optimizer

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.manual_seed(0)
model = build_model("ieee3modalPart", num_classes=171, loss="margin", pretrained=False, use_gpu=True, compute_dtype=torch.bfloat16, device=dev)
eng = Image3MEngine(_FakeDM(171), model, build_optimizer(model, optim="sgd", lr=1e-3, weight_decay=5e-4, momentum=0.9), margin=1,
                    weight_m=1, weight_x=1, use_gpu=True, label_smooth=True)
eng.defer_summary = True; eng.resident_batch = True
model.train()
batch = make_batch(64, 0, dev)


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showperflevel", "--showmaxpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
        return " | ".join(l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "Power", "Perf", "Temp")))[:900]
    except Exception as e:
        return "rocm-smi failed: %s" % e


def run(n, **env):
    for k, v in env.items():
        os.environ[k] = v
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n):
        eng.forward_backward(batch)
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


print("idle:", smi())
run(10)
print("plain (first 30 steps): %.2f ms" % run(30))
stop = []
th = threading.Thread(target=lambda: (time.sleep(0.5), stop.append(smi())))
th.start(); run(120); th.join()
print("under load:", stop[0])
dist.init_process_group("nccl", rank=0, world_size=1)
for ranges in model.grad_part_ranges():
    for a, b in ranges:
        dist.all_reduce(model._flat_grads[a:b])
run(25, IEEE_FORCE_DP_PATH="1")
for rnd in range(3):
    a = run(20, IEEE_FORCE_DP_PATH="0", IEEE_OPT_OVERLAP="1")
    b = run(20, IEEE_FORCE_DP_PATH="0", IEEE_OPT_OVERLAP="0")
    c = run(20, IEEE_FORCE_DP_PATH="1", IEEE_OPT_OVERLAP="1")
    print("round %d: plain %.2f ms | optimizer on the launch stream %.2f ms | staged (1-rank RCCL) %.2f ms" % (rnd, a, b, c))
os.environ["IEEE_FORCE_DP_PATH"] = "0"; os.environ["IEEE_OPT_OVERLAP"] = "1"
dist.destroy_process_group()
